"""lib/data_utils/fs.py of the reference: POSIX-like path/file helpers the eval scripts call
(walk, join, exists, dirname, open, read_bytes / aread_bytes -- the two the reference's own dataset
loader calls, lib/data_utils/fs.py:67-91).  `join` always treats the trailing parts as relative, like the
reference's (lib/data_utils/fs.py:25-52)."""
import io
import os

walk = os.walk
listdir = os.listdir
exists = os.path.exists
open = io.open  # noqa: A001


def _sep_of(root: str) -> str:
    i = max(root.rfind("/"), root.rfind("\\"))
    return root[i] if i >= 0 else "/"


def join(root, *parts):
    sep = _sep_of(root)
    tail = sep.join(p.strip("/\\") for p in parts if p.strip("/\\"))
    if not tail:
        return root
    return root + tail if root.endswith(sep) else root + sep + tail


def basename(path):
    return path[max(path.rfind("/"), path.rfind("\\")) + 1:]


def dirname(path):
    i = max(path.rfind("/"), path.rfind("\\"))
    return path[:i] if i > 0 else (path[: i + 1] if i == 0 else "")


def makedirs(p):
    os.makedirs(p, exist_ok=True)


def read_bytes(path, start=0, stop=None) -> bytes:
    """File contents, or the byte span [start, stop)."""
    with io.open(path, "rb") as f:
        if start:
            f.seek(start)
        return f.read() if stop is None else f.read(stop - start)


async def aread_bytes(path, start=0, stop=None) -> bytes:
    return read_bytes(path, start, stop)
