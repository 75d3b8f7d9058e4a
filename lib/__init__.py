"""Drop-in module paths of the reference (`lib.models`, `lib.tracker`, `lib.common`, `lib.data_utils`,
`lib.batched_dataset`): thin re-exports of absolutetrack_amd so that run_eval_known_skeleton.py /
run_eval_unknown_skeleton.py / run_inference_torch_data.py import the MI355X-native hot path unchanged
(SURVEY.md section 8 b).

Modules this package does not serve (`lib.data_utils.{async_dataset,async_utils,nested_async,dataset_util,
split}`: the asyncio dataset loader, out of scope) fall through to a same-named `lib/` directory later on
`sys.path` -- i.e. a reference checkout placed *behind* this repo on PYTHONPATH -- because every package here
extends its `__path__` (the reference's `lib/` is a namespace package without `__init__.py`).  Modules served
here always win: this directory is first on each `__path__`.  `sys.path` is read when `lib` is first imported."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
