"""Device-resident multi-frame pipeline and per-frame sharding across GPUs.

The reference processes frames one by one in a Python loop (scripts/test_vel_estimation.py:145-151,
plotting/movie_generator.py:138-150).  Frames are independent for every processor on the hot path
(SURVEY.md section 8e), so a batch is processed as ``[F, V, S, C]`` cubes resident in HBM, and a multi-GPU job
is a contiguous block split of the frame range -- one process per GPU, NO collective on the data path; only
the host-side join of per-frame results (``gather_frames``) touches ``torch.distributed``.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .detectors.ca_cfar import CaCFAR2D
from .processors.range_angle_resp import angle_tables


def shard_bounds(n_frames: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block split: frame f belongs to rank floor(f * world / n_frames)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    lo = -(-rank * n_frames // world)           # ceil(rank * F / world)
    hi = -(-(rank + 1) * n_frames // world)
    return lo, hi


def gather_frames(local: Sequence, n_frames: int, dist=None) -> Optional[List]:
    """Join per-frame results (one picklable object per local frame) on rank 0 in global frame order.

    ``dist`` is an initialised ``torch.distributed`` module (any backend with object collectives, e.g. gloo)
    or None for a single process.  Returns the list on rank 0, None elsewhere."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        if len(local) != n_frames:
            raise ValueError("single-process gather needs all frames")
        return list(local)
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_bounds(n_frames, rank, world)
    if len(local) != hi - lo:
        raise ValueError(f"rank {rank} holds {len(local)} frames, expected {hi - lo}")
    parts = [None] * world if rank == 0 else None
    dist.gather_object(list(local), parts, dst=0)
    if rank != 0:
        return None
    out: List = []
    for p in parts:
        out.extend(p)
    return out


def run_sharded(process_range: Callable[[int, int], Sequence], n_frames: int, dist=None) -> Optional[List]:
    """``process_range(lo, hi)`` -> per-frame results of this rank's block; joined on rank 0."""
    if dist is None or not dist.is_initialized():
        return list(process_range(0, n_frames))
    lo, hi = shard_bounds(n_frames, dist.get_rank(), dist.get_world_size())
    return gather_frames(process_range(lo, hi), n_frames, dist)


class FramePipeline:
    """``[F, V, S, C]`` complex64 cubes in HBM -> RD cube, detections, point clouds, 3-D FFT cube.

    Mirrors ``PointCloudGenerator(RangeDopplerDetector2D(CaCFAR2D/OsCFAR2D))`` and
    ``RangeAngleProcessorDBSEnhanced.compute_3d_windowed_fft`` per frame (reference:
    processors/point_cloud_generator.py:108-140, range_angle_resp_dbs_enhanced.py:137-198)."""

    def __init__(self, config_manager, max_frames: int, shape: Tuple[int, int, int], num_angle_bins: int = 64,
                 cfar=None, az_antenna_idxs=(), el_antenna_idxs=(), shift_az_resp=True, shift_el_resp=False,
                 det_capacity: int = 2048, ctx: _lib.Context = None):
        self.cm = config_manager
        self.V, self.S, self.C = (int(x) for x in shape)
        self.A = int(num_angle_bins)
        self.max_frames = int(max_frames)
        self.cfar = cfar if cfar is not None else CaCFAR2D((4, 4), (2, 2), 1e-5)
        self.az = [int(i) for i in az_antenna_idxs]
        self.el = [int(i) for i in el_antenna_idxs]
        self.shift_az, self.shift_el = bool(shift_az_resp), bool(shift_el_resp)
        self.cap = int(det_capacity)
        self.ctx = ctx if ctx is not None else _lib.default_context()
        self.bufs = _lib.BufferSet(self.ctx)
        self.n_frames = 0
        self.n_refined, self._l1_valid = 0, False
        cm = config_manager
        self.vel_bins = np.arange(-cm.vel_max_m_s, cm.vel_max_m_s - cm.vel_res_m_s + 1e-3, cm.vel_res_m_s)
        self.range_bins = np.arange(0, cm.range_max_m - cm.range_res_m / 2 + 1e-3, cm.range_res_m)
        _, self.angle_bins = angle_tables(self.A)
        self.cube_bytes = self.V * self.S * self.C * 8
        self.d_in = self.bufs.get("cubes", self.max_frames * self.cube_bytes)

    # ------------------------------------------------------------------ input
    def load(self, cubes: np.ndarray):
        cubes = np.ascontiguousarray(cubes, dtype=np.complex64)
        if cubes.ndim != 4 or cubes.shape[1:] != (self.V, self.S, self.C) or cubes.shape[0] > self.max_frames:
            raise ValueError(f"expected [F<={self.max_frames}, {self.V}, {self.S}, {self.C}] cubes, got {cubes.shape}")
        self.d_in.upload(cubes)
        self.n_frames = cubes.shape[0]

    def load_raw(self, raw_cubes: np.ndarray, num_tx: int):
        """``[F, num_rx, S, num_tx * loops]`` raw cubes (the layout ``VirtualArrayReformatter.process`` consumes,
        virtual_array_reformater.py:44-65): uploaded once, de-interleaved on the device into the virtual-array
        cubes every other method works on.  ``chain3d_raw`` skips even that pass."""
        raw = np.ascontiguousarray(raw_cubes, dtype=np.complex64)
        num_tx = int(num_tx)
        if raw.ndim != 4 or num_tx < 1 or self.V % num_tx or raw.shape[1] != self.V // num_tx or \
                raw.shape[2:] != (self.S, num_tx * self.C) or raw.shape[0] > self.max_frames:
            raise ValueError(f"expected [F<={self.max_frames}, {self.V}/num_tx, {self.S}, num_tx*{self.C}] raw cubes, "
                             f"got {raw.shape} with num_tx={num_tx}")
        self.d_raw = self.bufs.get("raw", self.max_frames * self.cube_bytes)
        self.d_raw.upload(raw)
        self.n_frames, self._raw_tx = raw.shape[0], num_tx
        _lib.check(self.ctx.lib.mmw_virtual_array_reformat(self.ctx.handle, self.d_raw.ptr, self.d_in.ptr, self.n_frames,
                                                           self.V // num_tx, num_tx, self.S, self.C))

    def chain3d_raw(self, magnitude: bool = False):
        """3-D windowed FFT straight from the raw cubes of ``load_raw`` (``mmw_chain3d_raw``)."""
        F, A, S, C = self.n_frames, self.A, self.S, self.C
        self.d_cube3d = self.bufs.get("cube3d", max(F, 1) * A * S * C * (4 if magnitude else 8))
        self._cube3d_mag = magnitude
        _lib.check(self.ctx.lib.mmw_chain3d_raw(self.ctx.handle, self.d_raw.ptr, None, self.d_cube3d.ptr, F,
                                                self.V // self._raw_tx, self._raw_tx, S, C, A, int(magnitude)))

    def synth(self, n_frames: int, seed0: int, num_targets: int = 8, noise_sigma: float = 30.0):
        if n_frames > self.max_frames:
            raise ValueError("n_frames exceeds max_frames")
        _lib.check(self.ctx.lib.mmw_synth_cubes(self.ctx.handle, self.d_in.ptr, n_frames, self.V, self.S, self.C,
                                                int(seed0), int(num_targets), float(noise_sigma)))
        self.n_frames = n_frames

    def cubes(self, lo: int = 0, hi: Optional[int] = None) -> np.ndarray:
        hi = self.n_frames if hi is None else hi
        return self.d_in.download((hi - lo, self.V, self.S, self.C), np.complex64, lo * self.cube_bytes)

    # ------------------------------------------------------------------ compute
    def chain3d(self, magnitude: bool = False):
        """3-D windowed FFT of every frame; result stays on the device (``fetch_chain3d`` copies frames out)."""
        F, A, S, C = self.n_frames, self.A, self.S, self.C
        esz = 4 if magnitude else 8
        self.d_cube3d = self.bufs.get("cube3d", max(F, 1) * A * S * C * esz)
        self._cube3d_mag = magnitude
        _lib.check(self.ctx.lib.mmw_chain3d(self.ctx.handle, self.d_in.ptr, None, self.d_cube3d.ptr, F, self.V, S, C, A,
                                            int(magnitude)))

    def fetch_chain3d(self, frame: int) -> np.ndarray:
        A, S, C = self.A, self.S, self.C
        if self._cube3d_mag:
            return self.d_cube3d.download((A, S, C), np.float32, frame * A * S * C * 4)
        return self.d_cube3d.download((A, S, C), np.complex64, frame * A * S * C * 8)

    def detect(self) -> List[np.ndarray]:
        """RD (all antennas, fp32) + float64 |RD| of antenna 0 + CFAR + ordered compaction for every frame.

        Returns the per-frame int64 ``(N, 2)`` [range_idx, doppler_idx] arrays (row-major order, == np.where)."""
        F, V, S, C, cap = self.n_frames, self.V, self.S, self.C, self.cap
        L, h = self.ctx.lib, self.ctx.handle
        n = S * C
        self.d_rd = self.bufs.get("rd", max(F, 1) * self.cube_bytes)
        d_mag = self.bufs.get("mag64", max(F, 1) * n * 8)
        d_mask = self.bufs.get("mask", max(F, 1) * n)
        self.d_dets = self.bufs.get("dets", max(F, 1) * cap * 8)
        self.d_cnt = self.bufs.get("counts", max(F, 1) * 4)
        (tr, td), (gr, gd) = self.cfar.num_train, self.cfar.num_guard
        for f0 in range(0, F, 32768):       # grid limits of the per-frame launches
            nf = min(32768, F - f0)
            _lib.check(L.mmw_detect_batch(h, self.d_in.at(f0 * self.cube_bytes), self.d_rd.at(f0 * self.cube_bytes),
                                          d_mag.at(f0 * n * 8), d_mask.at(f0 * n), self.d_dets.at(f0 * cap * 8),
                                          self.d_cnt.at(f0 * 4), nf, V, S, C, self.cfar.kind, int(tr), int(td), int(gr),
                                          int(gd), float(self.cfar._scale()), int(self.cfar._k_rank()), cap))
        self.counts = self.d_cnt.download((F,), np.int32)
        if np.any(self.counts > cap):
            raise _lib.MmwGpuError(f"detection capacity {cap} exceeded (max count {int(self.counts.max())}): "
                                   "raise det_capacity")
        dets = self.d_dets.download((F, cap, 2), np.int32)
        self.dets = [dets[f, :self.counts[f]].astype(np.int64) for f in range(F)]
        return self.dets

    def _argmax(self, ant, shift) -> np.ndarray:
        """Exact (float64-equivalent) argmax bins of every detection: ``mmw_angle_argmax_exact``."""
        F, cap = self.n_frames, self.cap
        d_idx = self.bufs.get("angle_idx", max(F, 1) * cap * 4)
        d_l1 = self.bufs.get("plane_l1", max(F, 1) * self.V * 4)
        L, h = self.ctx.lib, self.ctx.handle
        if not self._l1_valid:
            _lib.check(L.mmw_plane_l1(h, self.d_in.ptr, d_l1.ptr, F, self.V, self.S, self.C))
            self._l1_valid = True
        arr, n_ant = _lib.int_array(ant)
        n_ref = _lib.C.c_int(0)
        for f0 in range(0, F, 32768):
            nf = min(32768, F - f0)
            _lib.check(L.mmw_angle_argmax_exact(h, self.d_in.at(f0 * self.cube_bytes), d_l1.at(f0 * self.V * 4),
                                                self.d_rd.at(f0 * self.cube_bytes), self.d_dets.at(f0 * cap * 8),
                                                self.d_cnt.at(f0 * 4), d_idx.at(f0 * cap * 4), nf, self.V, self.S, self.C,
                                                cap, arr, n_ant, self.A, int(shift), _lib.C.byref(n_ref)))
            self.n_refined += n_ref.value
        return d_idx.download((F, cap), np.int32)

    def point_clouds(self) -> List[np.ndarray]:
        """Per-frame float64 ``(N, 4)`` (x, y, z, velocity), FLU frame (point_cloud_generator.py:216-248)."""
        dets = self.detect()
        self.n_refined, self._l1_valid = 0, False   # detections re-evaluated in float64 (near-ties of the float32 pass)
        az_idx = self._argmax(self.az, self.shift_az) if self.az else None
        el_idx = self._argmax(self.el, self.shift_el) if self.el else None
        out = []
        for f, d in enumerate(dets):
            n = d.shape[0]
            if n == 0:
                out.append(np.empty((0, 4)))
                continue
            az = self.angle_bins[az_idx[f, :n]] if az_idx is not None else np.zeros(n)
            el = self.angle_bins[el_idx[f, :n]] if el_idx is not None else np.zeros(n)
            rng, vel = self.range_bins[d[:, 0]], self.vel_bins[d[:, 1]]
            cos_el = np.cos(el)
            out.append(np.column_stack((rng * cos_el * np.cos(az), rng * cos_el * np.sin(az), rng * np.sin(el), vel)))
        self.az_idx = None if az_idx is None else [az_idx[f, :len(d)].astype(np.int64) for f, d in enumerate(dets)]
        self.el_idx = None if el_idx is None else [el_idx[f, :len(d)].astype(np.int64) for f, d in enumerate(dets)]
        return out
