"""Build libumetrack_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per translation unit under _build/ (recompiled when the source or any header is newer), compiled in
parallel, then one link."""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
OUT = os.path.join(HERE, "libumetrack_hip.so")
SOURCES = ["ut_api.hip", "conv_igemm.hip", "conv_patch.hip", "conv_block32.hip", "conv_split.hip", "conv_w4.hip", "conv_c64k.hip",
           "conv_c32s2.hip", "stem.hip", "head.hip", "fk.hip", "warp.hip", "cropgen.hip", "homography.hip", "metrics.hip"]
HEADERS = ["ut_kernels.h", "ut_math.h", "ut_fk.h", os.path.join("..", "..", "include", "umetrack_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _mtime(path: str) -> float:
    return os.path.getmtime(path) if os.path.exists(path) else 0.0


def _stale() -> bool:
    t = _mtime(OUT)
    return t == 0.0 or any(_mtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("UT_EXTRA_HIPCC_FLAGS", "").split()
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "flags.txt")
    flags_now = " ".join([hipcc] + FLAGS + extra)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    newest_header = max(_mtime(os.path.join(CSRC, f)) for f in HEADERS)
    jobs = []
    for f in SOURCES:
        src, obj = os.path.join(CSRC, f), os.path.join(OBJ, f + ".o")
        if force or _mtime(obj) < max(_mtime(src), newest_header):
            jobs.append([hipcc] + FLAGS + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + [os.path.join(OBJ, f + ".o") for f in SOURCES])
    with open(stamp, "w") as fh:
        fh.write(flags_now)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
