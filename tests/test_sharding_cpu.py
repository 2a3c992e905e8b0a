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


class _FakePart:
    """Host-only stand-in for FramePipeline: records what it was asked to do and answers from the frame seeds, so the
    split / join logic of MultiDeviceFramePipeline runs without a GPU."""

    def __init__(self, device, max_frames, shape):
        import threading
        self.device, self.max_frames, self.shape = device, max_frames, shape
        self.thread = threading.get_ident()
        self.threads_seen = set()
        self.frames = []
        self.n_refined = 0

    def _touch(self):
        import threading
        self.threads_seen.add(threading.get_ident())

    def load(self, cubes):
        self._touch()
        assert cubes.shape[0] <= self.max_frames
        self.frames = [int(round(c[0, 0, 0].real)) for c in cubes]      # the test stores the frame id in element 0

    def synth(self, n, seed0, **kw):
        self._touch()
        self.frames = list(range(seed0, seed0 + n))

    def cubes(self, lo, hi):
        self._touch()
        out = np.zeros((hi - lo,) + self.shape, dtype=np.complex64)
        for i, f in enumerate(self.frames[lo:hi]):
            out[i, 0, 0, 0] = f
        return out

    def detect(self):
        self._touch()
        self.dets = [np.array([[f, self.device]], dtype=np.int64) for f in self.frames]
        return self.dets

    def point_clouds(self):
        self.detect()
        self.n_refined = len(self.frames)
        return [np.full((1, 4), float(f)) for f in self.frames]

    def chain3d(self, magnitude=False):
        self._touch()

    def fetch_chain3d(self, local):
        self._touch()
        return np.full((2, 2, 2), self.frames[local], dtype=np.complex64)


@pytest.mark.parametrize("world,n_frames", [(1, 5), (2, 7), (4, 10), (8, 3), (3, 0)])
def test_multi_device_pipeline_split_and_join_with_fake_parts(world, n_frames):
    from mmwave_radar_processing_amd.batch import MultiDeviceFramePipeline
    shape = (2, 2, 2)
    made = []

    def factory(device, max_frames):
        made.append(_FakePart(device, max_frames, shape))
        return made[-1]
    mp = MultiDeviceFramePipeline(None, max_frames=16, shape=shape, devices=list(range(world)), part_factory=factory)
    assert [p.device for p in mp.parts] == list(range(world)) and all(p.max_frames == -(-16 // world) for p in mp.parts)
    cubes = np.zeros((n_frames,) + shape, dtype=np.complex64)
    cubes[:, 0, 0, 0] = np.arange(n_frames)
    mp.load(cubes)
    assert mp.bounds == [shard_bounds(n_frames, r, world) for r in range(world)]
    dets = mp.detect()
    assert [int(d[0, 0]) for d in dets] == list(range(n_frames))                   # frame order
    assert [int(d[0, 1]) for d in dets] == [f * world // n_frames for f in range(n_frames)]    # owner = floor(f W / F)
    pcs = mp.point_clouds()
    assert [float(p[0, 0]) for p in pcs] == [float(f) for f in range(n_frames)] and mp.n_refined == n_frames
    out = np.zeros((n_frames, 2, 2, 2), dtype=np.complex64)
    mp.chain3d(out=out)
    assert np.array_equal(out[:, 0, 0, 0].real, np.arange(n_frames))               # disjoint slices of a caller array
    np.testing.assert_array_equal(mp.cubes()[:, 0, 0, 0].real, np.arange(n_frames))
    for f in range(n_frames):
        assert mp.owner(f) == (f * world // n_frames, f - shard_bounds(n_frames, f * world // n_frames, world)[0])
        assert mp.fetch_chain3d(f)[0, 0, 0] == f
    mp.synth(n_frames, seed0=100)
    assert [int(d[0, 0]) for d in mp.detect()] == list(range(100, 100 + n_frames))  # seed0 + global frame index
    # every part is only ever touched by one thread, and no two parts share a thread
    threads = [p.threads_seen for p in mp.parts if p.threads_seen]
    assert all(len(t) == 1 for t in threads) and len(set.union(set(), *threads)) == len(threads)
    with pytest.raises(ValueError):
        mp.load(np.zeros((17,) + shape, dtype=np.complex64))
    mp.close()
