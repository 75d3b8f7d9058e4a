"""lib/models/umetrack_model.py of the reference -> absolutetrack_amd.model."""
from absolutetrack_amd.model import InputFrameData, InputFrameDesc, InputSkeletonData, UmeTrackModel  # noqa: F401
