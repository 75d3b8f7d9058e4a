"""lib/common/hand_skinning.py of the reference -> the HIP FK kernel (absolutetrack_amd.hand.skin_landmarks)."""
from absolutetrack_amd.hand import skin_landmarks  # noqa: F401
