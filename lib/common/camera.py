"""lib/common/camera.py of the reference -> absolutetrack_amd.geometry."""
from absolutetrack_amd.geometry import (  # noqa: F401
    CameraModel, Fisheye62CameraModel, NoDistortion, PinholePlaneCameraModel, read_camera_from_json)
