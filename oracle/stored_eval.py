"""Oracle tooling (TEST INFRASTRUCTURE ONLY): read the reference's stored evaluation
outputs sample_data/user05/recording_XX.npy WITHOUT unpickling them.

Those files are pickled dicts {tracked_keypoints, gt_keypoints, valid_tracking}
written by run_eval_known_skeleton.py:96-104.  numpy's safe loader refuses them
("This file contains pickled (object) data").  Nothing from the file may be
executed, so this reader only TOKENISES the byte stream with
`pickletools.genops` (which never builds objects or calls REDUCE/GLOBAL targets) and
lifts out, for each dict key, the raw little-endian payload bytes, the dtype code
string and the shape tuple that appear as literal opcodes arguments.
"""
import pickletools
from typing import Dict

import numpy as np

_DTYPES = {"f8": np.float64, "f4": np.float32, "b1": np.bool_, "i8": np.int64, "i4": np.int32}


def read_stored_eval(path: str) -> Dict[str, np.ndarray]:
    data = open(path, "rb").read()
    out: Dict[str, np.ndarray] = {}
    key = None
    ints = []          # integer literals seen since the last MARK (shape candidates)
    shape = None
    dtype = None
    last_dtype = None  # dtype objects are memoised and re-used via BINGET by later arrays
    for op, arg, _pos in pickletools.genops(data):
        name = op.name
        if name in ("SHORT_BINUNICODE", "BINUNICODE"):
            if arg in ("tracked_keypoints", "gt_keypoints", "valid_tracking"):
                key, shape, dtype = arg, None, None
            elif arg in _DTYPES:
                dtype = last_dtype = _DTYPES[arg]
            continue
        if name == "MARK":
            ints = []
            continue
        if name in ("BININT", "BININT1", "BININT2"):
            ints.append(int(arg))
            continue
        if name in ("TUPLE", "TUPLE1", "TUPLE2", "TUPLE3"):
            # the first all-positive int tuple after the key and before the payload is the shape
            take = {"TUPLE": len(ints), "TUPLE1": 1, "TUPLE2": 2, "TUPLE3": 3}[name]
            cand = ints[len(ints) - take:] if take <= len(ints) else []
            if key is not None and shape is None and len(cand) >= 2 and all(i > 0 for i in cand):
                shape = tuple(cand)
            ints = []
            continue
        if name in ("BINBYTES", "SHORT_BINBYTES", "BINBYTES8") and key is not None and len(arg) > 8:
            dt = dtype or last_dtype
            arr = np.frombuffer(arg, dtype=dt)
            if shape is None or int(np.prod(shape)) != arr.size:
                raise ValueError(f"{path}: could not recover shape for {key}")
            out[key] = arr.reshape(shape).copy()
            key = None
            continue
    # valid_tracking re-uses nothing; gt_keypoints re-uses the memoised f8 dtype of tracked_keypoints
    missing = {"tracked_keypoints", "gt_keypoints", "valid_tracking"} - set(out)
    if missing:
        raise ValueError(f"{path}: missing {missing}")
    return out
