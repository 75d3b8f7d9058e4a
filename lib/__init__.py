"""Drop-in module paths of the reference (`lib.models`, `lib.tracker`, `lib.common`, `lib.data_utils`):
thin re-exports of absolutetrack_amd so that run_eval_known_skeleton.py / run_eval_unknown_skeleton.py /
run_inference_torch_data.py import the MI355X-native hot path unchanged (SURVEY.md section 8 b)."""
