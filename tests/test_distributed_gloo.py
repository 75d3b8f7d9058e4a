"""world_size-2 `gloo` test of the multi-GPU path on CPU: contiguous frame sharding + all-gather of the
per-hand records (the only communication of the path; RCCL on the GPU node, gloo here).  The per-rank
compute is a deterministic stand-in (the HIP kernels cannot run on CPU); what is checked is that the
gathered tensor is the rank-ordered concatenation of the shards, i.e. frame order is preserved."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    from absolutetrack_amd import pipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = pipeline.shard_frames(n_frames, rank, world)
    # stand-in records: 2 hands per frame, record[k] = global hand-frame index + k/1000
    s_local = 2 * (hi - lo)
    idx = torch.arange(2 * lo, 2 * hi, dtype=torch.float32)
    rec = idx[:, None] + torch.arange(pipeline.RECORD, dtype=torch.float32)[None] / 1000.0
    assert rec.shape == (s_local, pipeline.RECORD)
    out = pipeline.gather_records(rec, world)
    q.put((rank, out[:, 0].tolist(), tuple(out.shape)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_preserves_frame_order_world2():
    world, n_frames = 2, 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _rank, col0, shape in results:
        assert shape == (2 * n_frames, 123)
        assert col0 == [float(i) for i in range(2 * n_frames)]     # every rank holds all records, in frame order


def test_gather_records_single_rank_is_identity():
    sys.path.insert(0, ROOT)
    from absolutetrack_amd import pipeline
    x = torch.randn(5, pipeline.RECORD)
    assert pipeline.gather_records(x, 1) is x
