"""lib/tracker/tracking_result.py of the reference -> absolutetrack_amd.tracker."""
from absolutetrack_amd.tracker import SingleHandPose, TrackingResult  # noqa: F401
