#!/usr/bin/env python3
"""Diagnostic: do hipMemsetAsync nodes captured into a hipGraph run, and in order, on replay?  (ROCm 7.2)"""
import ctypes
import faulthandler
import torch

faulthandler.dump_traceback_later(60, exit=True)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda", 0)
for nbytes, off in ((512, 0), (4, 4), (16384, 0), (32768, 0)):
    x = torch.full((16384,), 7, dtype=torch.int32, device=dev)
    y = torch.zeros(4, dtype=torch.int32, device=dev)
    n = nbytes // 4
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            x[off // 4: off // 4 + n].add_(1)                      # a kernel in front of the memset
            rc = hip.hipMemsetAsync(ctypes.c_void_p(x.data_ptr() + off), 0, nbytes, st)
            x[off // 4: off // 4 + n].add_(1)                      # and one behind it: 1 if the memset ran in between
            y[0] = x[off // 4]
            y[1] = x[off // 4 + n - 1]
            y[2] = x[off // 4 + n] if off // 4 + n < x.numel() else 0    # the word after the range: untouched (7)
            y[3] = x[0]
    out = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        out.append(y.tolist())
    print(f"memset {nbytes} B at +{off}: rc={rc} replays -> {out}   (want [1, 1, 7, .] each time)", flush=True)
