"""lib/data_utils/idxbinfile.py of the reference (.torch.idx / .torch.bin) -> absolutetrack_amd.formats."""
from absolutetrack_amd.formats import IDX_MAGIC, TorchIdx, write_torch_idx_bin  # noqa: F401
