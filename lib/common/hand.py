"""lib/common/hand.py of the reference -> absolutetrack_amd.hand."""
from absolutetrack_amd.hand import (  # noqa: F401
    DOF_PER_FINGER, LANDMARK, LEFT_HAND_INDEX, NUM_DIGITS, NUM_FINGERTIPS_PER_HAND, NUM_HANDS, NUM_JOINT_FRAMES,
    NUM_JOINTS_PER_HAND, NUM_LANDMARKS_PER_HAND, RIGHT_HAND_INDEX, HandModel, mirrored_hand_model, scaled_hand_model)
