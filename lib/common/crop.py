"""lib/common/crop.py of the reference -> absolutetrack_amd.geometry."""
from absolutetrack_amd.geometry import gen_crop_parameters_from_points, gen_intrinsics_from_bounding_pts  # noqa: F401
