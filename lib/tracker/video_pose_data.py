"""lib/tracker/video_pose_data.py of the reference -> absolutetrack_amd.formats (label JSON, view split, stream).
mp4 decoding needs PyAV, which is outside the scope of this package: VideoStream raises ImportError without it."""
from absolutetrack_amd.formats import (HandPoseLabels, SyncedImagePoseStream, VideoStream, _load_hand_pose_labels,  # noqa: F401
                                       _load_json, load_hand_model_from_dict)
