"""lib/batched_dataset/sample.py of the reference -> absolutetrack_amd.torch_data."""
from absolutetrack_amd.torch_data import RawSample, parse_raw_buffers  # noqa: F401
