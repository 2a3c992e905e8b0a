"""N>1 path on CPU: block split of the frame range + host-side join over torch.distributed (gloo, world_size 2).
The per-frame worker here is the CPU oracle (tests may use it); on the GPU box the same split/join wraps
FramePipeline (tests/test_gpu_parity.py::test_frame_pipeline_*)."""
import os
import socket

import numpy as np
import pytest

from mmwave_radar_processing_amd import synth
from mmwave_radar_processing_amd.batch import gather_frames, run_sharded, shard_bounds
from oracle import oracle_np as O


def test_shard_bounds_partition_every_frame_once():
    for n in (0, 1, 7, 8, 10, 1250, 10000):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
            for r, (a, b) in enumerate(spans):     # frame f -> rank floor(f * world / n)
                for f in (a, b - 1):
                    if a <= f < b:
                        assert f * world // n == r
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)
    assert shard_bounds(10000, 3, 8) == (3750, 5000)


def _frame_dets(f):
    cube = synth.synth_cube(500 + f, (4, 32, 16), num_targets=3)
    _, _, dets, _, _ = O.rd_detect_2d(cube, (2, 2), (1, 1), 1e-3)
    return dets


def _worker(rank, world, port, n_frames, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        seen = []

        def process_range(lo, hi):
            seen.append((lo, hi))
            return [_frame_dets(f) for f in range(lo, hi)]
        out = run_sharded(process_range, n_frames, dist)
        # bench-style control plane: barrier, then MAX of a per-rank time
        import torch
        dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, seen[0], None if out is None else [o.tolist() for o in out], float(t.item())))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_sharded_run_matches_serial():
    import multiprocessing as mp
    n_frames, world = 7, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_frames, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(world):
        rank, span, out, tmax = q.get(timeout=240)
        results[rank] = (span, out, tmax)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][0] == (0, 4) and results[1][0] == (4, 7)
    assert results[1][1] is None and results[0][2] == 2.0 and results[1][2] == 2.0
    serial = [_frame_dets(f).tolist() for f in range(n_frames)]
    assert results[0][1] == serial
    assert any(len(d) > 0 for d in serial)


def test_single_process_join_is_identity():
    items = [np.arange(i) for i in range(5)]
    assert [x.tolist() for x in gather_frames(items, 5)] == [x.tolist() for x in items]
    assert run_sharded(lambda lo, hi: list(range(lo, hi)), 6) == list(range(6))
    with pytest.raises(ValueError):
        gather_frames(items, 6)
