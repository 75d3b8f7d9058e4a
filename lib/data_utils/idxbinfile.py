"""lib/data_utils/idxbinfile.py of the reference (.torch.idx / .torch.bin) -> absolutetrack_amd.formats."""
from absolutetrack_amd.formats import (  # noqa: F401
    IDX_MAGIC, OBJECT_DTYPE, BinFormat, Buffer, MsgpackObject, RawField, TorchIdx, write_torch_idx_bin)
