"""world_size 2 and 3 `gloo` tests of the multi-GPU path on CPU: contiguous frame sharding + all-gather of the
per-hand records (the only communication of the path; RCCL on the GPU node, gloo here).  The per-rank
compute is a deterministic stand-in (the HIP kernels cannot run on CPU); what is checked is that the
gathered tensor is the rank-ordered concatenation of the shards, i.e. frame order is preserved - also when the
frame count does not divide by the world size and when a rank lost a hand to the confidence / visibility gate
(SURVEY.md section 8 e; the reference's only distribution idiom is lib/data_utils/async_dataset.py:546-559)."""
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


def _worker(rank, world, port, n_frames, q, dropped=(), equal_counts=False):
    sys.path.insert(0, ROOT)
    from absolutetrack_amd import pipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = pipeline.shard_frames(n_frames, rank, world)
    # stand-in records: 2 hands per frame minus the dropped ones, record[k] = global hand-frame index + k/1000
    idx = torch.tensor([i for i in range(2 * lo, 2 * hi) if i not in dropped], dtype=torch.float32)
    rec = idx[:, None] + torch.arange(pipeline.RECORD, dtype=torch.float32)[None] / 1000.0
    try:
        out = pipeline.gather_records(rec, world, equal_counts=equal_counts)
        q.put((rank, out[:, 0].tolist(), tuple(out.shape), bool(torch.equal(out[:, 1], out[:, 0] + 0.001))))
    except Exception as e:                   # noqa: BLE001 - reported to the parent
        q.put((rank, repr(e), None, False))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n_frames, dropped=(), equal_counts=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q, dropped, equal_counts)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def test_gather_records_preserves_frame_order_world2():
    world, n_frames = 2, 12
    for equal_counts in (False, True):
        for _rank, col0, shape, ok in _run(world, n_frames, equal_counts=equal_counts):
            assert shape == (2 * n_frames, 123) and ok
            assert col0 == [float(i) for i in range(2 * n_frames)]     # every rank holds all records, in frame order


def test_gather_records_unequal_shards():
    """F % world != 0 (blocks differ by one frame) at world 2 and 3, and one hand dropped on one rank."""
    for world, n_frames, dropped in ((2, 7, ()), (3, 10, ()), (3, 11, (5,)), (2, 6, (0, 11)), (3, 2, ())):
        want = [float(i) for i in range(2 * n_frames) if i not in dropped]
        for _rank, col0, shape, ok in _run(world, n_frames, dropped):
            assert shape == (len(want), 123) and ok, (world, n_frames, dropped, col0)
            assert col0 == want


def test_gather_records_single_rank_is_identity():
    sys.path.insert(0, ROOT)
    from absolutetrack_amd import pipeline
    x = torch.randn(5, pipeline.RECORD)
    assert pipeline.gather_records(x, 1) is x


def _seq_worker(rank, world, port, n_seq, n_steps, q):
    """Stand-in for sequence mode: the temporal state is a per-slot recurrence state[slot] = 0.5 * state[slot] + x(seq, hand, t)
    kept rank-locally and indexed by the descriptors of pipeline.sequence_step_descriptors; a record is the state after
    the step.  What is checked is that sharding by sequence + gathering per step reproduces the single-process run."""
    sys.path.insert(0, ROOT)
    from absolutetrack_amd import pipeline
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = pipeline.shard_sequences(n_seq, rank, world)
    seq = torch.arange(lo, hi).repeat_interleave(2)                       # two hands per sequence, sequence-major
    hand = torch.tensor([0, 1] * (hi - lo), dtype=torch.long)
    state = torch.zeros(0)
    out = []
    for t in range(n_steps):
        mem_idx, use, n_slots = pipeline.sequence_step_descriptors(hand, first_step=t == 0)
        if state.shape[0] < n_slots:
            state = torch.cat([state, torch.zeros(n_slots - state.shape[0])])
        x = (seq * 2 + hand).float() * 0.25 + t                            # the step's input of (sequence, hand)
        prev = torch.where(use.bool(), state[mem_idx], torch.zeros(len(mem_idx)))
        state[mem_idx] = 0.5 * prev + x
        rec = state[mem_idx][:, None] + torch.arange(pipeline.RECORD, dtype=torch.float32)[None] / 1000.0
        out.append(pipeline.gather_records(rec, world)[:, 0].clone())
    q.put((rank, torch.stack(out).tolist()))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_sequence_sharding_matches_single_process():
    """pipeline.shard_sequences: whole hand-sequences per rank (temporal slots rank-local), per-step all-gather in
    sequence order; 5 sequences x 2 hands x 4 steps at world 2 and 3 (unequal blocks) against world 1."""
    ctx = mp.get_context("spawn")
    n_seq, n_steps = 5, 4

    def run(world):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_seq_worker, args=(r, world, port, n_seq, n_steps, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = [q.get(timeout=180) for _ in range(world)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        return res

    want = run(1)[0][1]
    assert len(want) == n_steps and len(want[0]) == 2 * n_seq
    assert want[1] != want[0]                                              # the state is engaged
    for world in (2, 3):
        for _rank, got in run(world):
            assert got == want, world
