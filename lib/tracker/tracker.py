"""lib/tracker/tracker.py of the reference -> absolutetrack_amd.tracker."""
from absolutetrack_amd.tracker import (  # noqa: F401
    CONFIDENCE_THRESHOLD, M_TO_MM, MAX_VIEW_NUM, MIN_OBSERVED_LANDMARKS, MM_TO_M, HandTracker, HandTrackerOpts,
    InputFrame, ViewData)
