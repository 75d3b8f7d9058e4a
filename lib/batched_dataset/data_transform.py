"""lib/batched_dataset/data_transform.py of the reference -> absolutetrack_amd.torch_data (crop matrices and the
homography resampler run on the GPU: ut_gen_crop_matrices, ut_resample_homography)."""
from absolutetrack_amd.torch_data import (ModelInput, ModelTarget, PerBranchOutput, PoseData,  # noqa: F401
                                          _perspective_crop_images, prepare_inputs_targets, preprocess, scalar_type)
