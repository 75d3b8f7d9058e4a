"""lib/common/metric_utils.py of the reference -> absolutetrack_amd.metrics."""
from absolutetrack_amd.metrics import (MAX_LANDMARK_ERROR_MM, PCK_THRESHOLDS, PCK_curve, _PCK_curve, _safe_div,  # noqa: F401
                                       normalized_AUC)
