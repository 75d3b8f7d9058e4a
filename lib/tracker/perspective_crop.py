"""lib/tracker/perspective_crop.py of the reference -> absolutetrack_amd.tracker."""
from absolutetrack_amd.tracker import (  # noqa: F401
    gen_crop_cameras_from_pose, landmarks_from_hand_pose, neutral_joint_angles, rank_hand_visibility_in_cameras,
    skin_landmarks_np)
