"""Build libumetrack_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libumetrack_hip.so")
SOURCES = ["ut_api.hip", "conv_igemm.hip", "conv_patch.hip", "conv_block32.hip", "conv_split.hip", "conv_c64r.hip", "conv_c64k.hip", "conv_c32s2.hip", "stem.hip", "head.hip", "fk.hip", "warp.hip", "cropgen.hip", "homography.hip", "metrics.hip"]
HEADERS = ["ut_kernels.h", "ut_math.h", "ut_fk.h", os.path.join("..", "..", "include", "umetrack_hip.h")]


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall",
           "-Wno-unused-function", "-o", OUT] + os.environ.get("UT_EXTRA_HIPCC_FLAGS", "").split() \
        + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
