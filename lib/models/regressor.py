"""lib/models/regressor.py of the reference: only the output container crosses the boundary."""
from absolutetrack_amd.model import RegressorOutput  # noqa: F401
