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
        self.n_refined = 0
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
        self.n_frames, self._raw_tx, self.d_raw_i16 = raw.shape[0], num_tx, None
        _lib.check(self.ctx.lib.mmw_virtual_array_reformat(self.ctx.handle, self.d_raw.ptr, self.d_in.ptr, self.n_frames,
                                                           self.V // num_tx, num_tx, self.S, self.C))

    def load_raw_i16(self, raw_iq: np.ndarray, num_tx: int):
        """``[F, num_rx, S, num_tx * loops, 2]`` int16 I/Q samples: uploaded as they are (half the bytes of complex64)
        and converted + de-interleaved on the device.  No upstream oracle for this layout (the reference receives complex
        cubes from a dataset reader that is not in its tree): it is the raw cube of ``load_raw`` with int16 pairs."""
        raw = np.ascontiguousarray(raw_iq, dtype=np.int16)
        num_tx = int(num_tx)
        if raw.ndim != 5 or raw.shape[4] != 2 or num_tx < 1 or self.V % num_tx or raw.shape[1] != self.V // num_tx or \
                raw.shape[2:4] != (self.S, num_tx * self.C) or raw.shape[0] > self.max_frames:
            raise ValueError(f"expected [F<={self.max_frames}, {self.V}/num_tx, {self.S}, num_tx*{self.C}, 2] int16 samples, "
                             f"got {raw.shape} with num_tx={num_tx}")
        d_i16 = self.bufs.get("raw_i16", self.max_frames * self.cube_bytes // 2)
        d_i16.upload(raw)
        self.n_frames, self._raw_tx, self.d_raw_i16 = raw.shape[0], num_tx, d_i16
        self.d_raw = None
        _lib.check(self.ctx.lib.mmw_virtual_array_reformat_i16(self.ctx.handle, d_i16.ptr, self.d_in.ptr, self.n_frames,
                                                               self.V // num_tx, num_tx, self.S, self.C))

    def stream(self, chunks, work: Callable[["FramePipeline"], object] = None, num_tx: int = 0, pinned: bool = False):
        """Host-resident frame loop (the reference's scripts/test_vel_estimation.py:145-151): iterate ``chunks`` of host cubes
        and yield ``work(self)`` per chunk (default: ``point_clouds()``), with the upload of chunk k + 1 on the copy queue
        while chunk k is processed.

        ``chunks``: ``[F_k <= max_frames, V, S, C]`` complex64 cubes, or -- with ``num_tx > 0`` -- int16 I/Q raw cubes
        ``[F_k, V / num_tx, S, num_tx * C, 2]`` (layout of ``load_raw_i16``: no upstream oracle).  ``pinned=True``: the chunks
        already live in pinned memory (``ctx.host_array``) and are copied from where they are; otherwise each chunk is first
        copied into one of two pinned staging blocks (a host memcpy, usually the slowest stage of the loop)."""
        ctx, Q = self.ctx, _lib
        work = work or (lambda p: p.point_clouds())
        i16 = num_tx > 0
        frame_bytes = self.cube_bytes // 2 if i16 else self.cube_bytes
        dev = [self.bufs.get("stream_in0", self.max_frames * frame_bytes), self.bufs.get("stream_in1", self.max_frames * frame_bytes)]
        ev_up, ev_free = [ctx.event(), ctx.event()], [ctx.event(), ctx.event()]
        # pinned staging blocks live with the pipeline (grow-only, like its device buffers): a per-file loop calling stream()
        # again and again re-uses them instead of pinning max_frames * frame_bytes twice per call
        stage = self.__dict__.setdefault("_stage", [None, None])
        d_cubes_own = self.d_in
        saved_raw = (getattr(self, "d_raw", None), getattr(self, "d_raw_i16", None), getattr(self, "_raw_tx", 0))
        it = iter(chunks)

        def upload(k, chunk):
            b = k % 2
            a = np.asarray(chunk)
            want = np.int16 if i16 else np.complex64
            if a.dtype != want or not a.flags.c_contiguous:
                a = np.ascontiguousarray(a, dtype=want)
            n = a.shape[0]
            if n > self.max_frames or a.nbytes != n * frame_bytes:
                raise ValueError(f"chunk of shape {a.shape} does not hold <= {self.max_frames} frames of {frame_bytes} bytes")
            if k >= 2:
                ctx.wait(Q.QUEUE_COPY, ev_free[b])          # the compute of chunk k - 2 has read device block b
            src = a
            if not pinned:
                if k >= 2:
                    ctx.event_sync(ev_up[b])                 # copy k - 2 has left staging block b
                if stage[b] is None or stage[b].nbytes < self.max_frames * frame_bytes:
                    if stage[b] is not None:
                        ctx.host_free(stage[b])
                    stage[b] = ctx.host_array((self.max_frames * frame_bytes,), np.uint8)
                src = stage[b][:a.nbytes]
                src[:] = a.reshape(-1).view(np.uint8)
            ctx.copy_async(dev[b].ptr, src.ctypes.data, a.nbytes, to_host=False, queue=Q.QUEUE_COPY)
            ctx.record(ev_up[b], Q.QUEUE_COPY)
            return n, src

        nxt = next(it, None)
        k = 0
        pending = upload(0, nxt) if nxt is not None else None
        keep = []                                           # sources of copies in flight stay referenced
        try:
            while pending is not None:
                n, src = pending
                keep = [src]
                nxt = next(it, None)
                pending = upload(k + 1, nxt) if nxt is not None else None
                if pending is not None:
                    keep.append(pending[1])
                b = k % 2
                ctx.wait(Q.QUEUE_COMPUTE, ev_up[b])
                self.n_frames = n
                if i16:
                    self.d_in = d_cubes_own
                    _lib.check(ctx.lib.mmw_virtual_array_reformat_i16(ctx.handle, dev[b].ptr, self.d_in.ptr, n, self.V // num_tx,
                                                                      num_tx, self.S, self.C))
                    # work() may call chain3d_raw(): it reads THIS chunk's raw samples
                    self.d_raw, self.d_raw_i16, self._raw_tx = None, dev[b], num_tx
                else:
                    self.d_in = dev[b]
                    self.d_raw = self.d_raw_i16 = None     # no raw cube behind a streamed virtual-array chunk
                out = work(self)
                ctx.record(ev_free[b], Q.QUEUE_COMPUTE)
                yield out
                k += 1
        finally:
            self.d_in = d_cubes_own
            self.d_raw, self.d_raw_i16, self._raw_tx = saved_raw
            ctx.sync()
            for e in ev_up + ev_free:
                _lib.check(ctx.lib.mmw_event_destroy(ctx.handle, e))

    def chain3d_raw(self, magnitude: bool = False):
        """3-D windowed FFT straight from the raw cubes of ``load_raw`` / ``load_raw_i16`` (``mmw_chain3d_raw[_i16]``)."""
        F, A, S, C = self.n_frames, self.A, self.S, self.C
        self.d_cube3d = self.bufs.get("cube3d", max(F, 1) * A * S * C * (4 if magnitude else 8))
        self._cube3d_mag = magnitude
        if getattr(self, "d_raw_i16", None) is None and getattr(self, "d_raw", None) is None:
            raise ValueError("chain3d_raw() needs the raw cubes of load_raw() / load_raw_i16() (or an int16 stream() chunk)")
        if getattr(self, "d_raw_i16", None) is not None:        # int16 cells: converted inside the first kernel's loads
            _lib.check(self.ctx.lib.mmw_chain3d_raw_i16(self.ctx.handle, self.d_raw_i16.ptr, None, self.d_cube3d.ptr, F,
                                                        self.V // self._raw_tx, self._raw_tx, S, C, A, int(magnitude)))
            return
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

    def _alloc_detect(self):
        F, V, cap = self.n_frames, self.V, self.cap
        self.d_rd = self.bufs.get("rd", max(F, 1) * self.cube_bytes)
        self.d_dets = self.bufs.get("dets", max(F, 1) * cap * 8)
        self.d_cnt = self.bufs.get("counts", max(F, 1) * 4)
        self.d_l1 = self.bufs.get("plane_l1", max(F, 1) * V * 4)     # error-bound scale (CFAR screening, exact argmax)

    def _cfar_args(self):
        (tr, td), (gr, gd) = self.cfar.num_train, self.cfar.num_guard
        return self.cfar.kind, int(tr), int(td), int(gr), int(gd), float(self.cfar._scale()), int(self.cfar._k_rank())

    def _detect_float64(self, f0: int, nf: int):
        """The float64 path for frames [f0, f0 + nf): RD + float64 |RD| of antenna 0 + CFAR + ordered compaction."""
        V, S, C, cap = self.V, self.S, self.C, self.cap
        n = S * C
        d_mag = self.bufs.get("mag64", max(self.n_frames, 1) * n * 8)
        d_mask = self.bufs.get("mask", max(self.n_frames, 1) * n)
        kind, tr, td, gr, gd, scale, k_rank = self._cfar_args()
        _lib.check(self.ctx.lib.mmw_detect_batch(self.ctx.handle, self.d_in.at(f0 * self.cube_bytes), self.d_rd.at(f0 * self.cube_bytes),
                                                 d_mag.at(f0 * n * 8), d_mask.at(f0 * n), self.d_dets.at(f0 * cap * 8),
                                                 self.d_cnt.at(f0 * 4), self.d_l1.at(f0 * V * 4), nf, V, S, C, kind, tr, td, gr, gd,
                                                 scale, k_rank, cap))

    def _argmax_float64(self, d_idx, ant, shift, f0: int, nf: int):
        cap = self.cap
        arr, n_ant = _lib.int_array(ant)
        n_ref = _lib.C.c_int(0)
        _lib.check(self.ctx.lib.mmw_angle_argmax_exact(self.ctx.handle, self.d_in.at(f0 * self.cube_bytes), self.d_l1.at(f0 * self.V * 4),
                                                       self.d_rd.at(f0 * self.cube_bytes), self.d_dets.at(f0 * cap * 8),
                                                       self.d_cnt.at(f0 * 4), d_idx.at(f0 * cap * 4), nf, self.V, self.S, self.C,
                                                       cap, arr, n_ant, self.A, int(shift), _lib.C.byref(n_ref)))
        self.n_refined += n_ref.value

    def _fused_supported(self, with_angles: bool) -> bool:
        kind, tr, td, gr, gd, _, _ = self._cfar_args()
        n_az, n_el = (len(self.az), len(self.el)) if with_angles else (0, 0)
        return bool(self.ctx.lib.mmw_detect_points_supported(self.S, self.C, kind, tr, td, gr, gd, n_az, n_el, self.A))

    def _detect_fused(self, with_angles: bool):
        """``mmw_detect_points``: RD + screened CFAR (undecided cells settled in float64) + ordered compaction (+ the
        azimuth / elevation argmax bins) in one pass; frames it hands back (count -1) go through the float64 path."""
        F, V, S, C, cap = self.n_frames, self.V, self.S, self.C, self.cap
        kind, tr, td, gr, gd, scale, k_rank = self._cfar_args()
        az, n_az = _lib.int_array(self.az if with_angles else [])
        el, n_el = _lib.int_array(self.el if with_angles else [])
        self.d_az = self.bufs.get("az_idx", max(F, 1) * cap * 4) if n_az else None
        self.d_el = self.bufs.get("el_idx", max(F, 1) * cap * 4) if n_el else None
        stats = (_lib.C.c_int * 5)()
        self.screen_stats = np.zeros(5, dtype=np.int64)
        step = max(1, min(F, (2 ** 31 - 1) // max(cap, 1)))
        for f0 in range(0, F, step):
            nf = min(step, F - f0)
            _lib.check(self.ctx.lib.mmw_detect_points(
                self.ctx.handle, self.d_in.at(f0 * self.cube_bytes), self.d_rd.at(f0 * self.cube_bytes), self.d_l1.at(f0 * V * 4),
                None, self.d_dets.at(f0 * cap * 8), self.d_cnt.at(f0 * 4), self.d_az.at(f0 * cap * 4) if n_az else None,
                self.d_el.at(f0 * cap * 4) if n_el else None, nf, V, S, C, kind, tr, td, gr, gd, scale, k_rank, cap,
                az, n_az, int(self.shift_az), el, n_el, int(self.shift_el), self.A, stats))
            self.screen_stats += np.array(list(stats), dtype=np.int64)
        self.n_refined += int(self.screen_stats[3] + self.screen_stats[4])
        counts = self.d_cnt.download((F,), np.int32)
        for f in np.nonzero(counts < 0)[0]:            # not decidable by the screening pass: the float64 path, frame by frame
            self._detect_float64(int(f), 1)
            if n_az:
                self._argmax_float64(self.d_az, self.az, self.shift_az, int(f), 1)
            if n_el:
                self._argmax_float64(self.d_el, self.el, self.shift_el, int(f), 1)
        if np.any(counts < 0):
            counts = self.d_cnt.download((F,), np.int32)
        return counts

    def _fetch_dets(self, counts) -> List[np.ndarray]:
        F, cap = self.n_frames, self.cap
        self.counts = counts
        if np.any(counts > cap):
            raise _lib.MmwGpuError(f"detection capacity {cap} exceeded (max count {int(counts.max())}): "
                                   "raise det_capacity")
        dets = self.d_dets.download((F, cap, 2), np.int32)
        self.dets = [dets[f, :counts[f]].astype(np.int64) for f in range(F)]
        return self.dets

    def detect(self) -> List[np.ndarray]:
        """RD (all antennas, fp32) + CFAR on antenna 0 + ordered compaction for every frame.

        Returns the per-frame int64 ``(N, 2)`` [range_idx, doppler_idx] arrays (row-major order, == np.where)."""
        F = self.n_frames
        self._alloc_detect()
        if self._fused_supported(False):
            return self._fetch_dets(self._detect_fused(False))
        for f0 in range(0, F, 32768):       # grid limits of the per-frame launches
            self._detect_float64(f0, min(32768, F - f0))
        return self._fetch_dets(self.d_cnt.download((F,), np.int32))

    def _argmax(self, ant, shift, name) -> np.ndarray:
        """Exact (float64-equivalent) argmax bins of every detection: ``mmw_angle_argmax_exact``."""
        F, cap = self.n_frames, self.cap
        d_idx = self.bufs.get(name, max(F, 1) * cap * 4)
        for f0 in range(0, F, 32768):
            self._argmax_float64(d_idx, ant, shift, f0, min(32768, F - f0))
        return d_idx.download((F, cap), np.int32)

    def point_clouds(self) -> List[np.ndarray]:
        """Per-frame float64 ``(N, 4)`` (x, y, z, velocity), FLU frame (point_cloud_generator.py:216-248)."""
        self.n_refined = 0      # detections re-evaluated in float64 (near-ties of the float32 pass)
        F, cap = self.n_frames, self.cap
        self._alloc_detect()
        if self._fused_supported(True):
            dets = self._fetch_dets(self._detect_fused(True))
            az_idx = self.d_az.download((F, cap), np.int32) if self.az else None
            el_idx = self.d_el.download((F, cap), np.int32) if self.el else None
        else:
            dets = self.detect()
            az_idx = self._argmax(self.az, self.shift_az, "az_idx") if self.az else None
            el_idx = self._argmax(self.el, self.shift_el, "el_idx") if self.el else None
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


class MultiDeviceFramePipeline:
    """One process, every visible device: the frame range is block-split (``shard_bounds``) over one ``FramePipeline``
    per device, each driven by its own host thread (a context is only ever touched by its thread), results land in
    disjoint slices of caller-owned arrays and per-frame lists come back concatenated in frame order.  No device
    talks to another (no collective).  This is what the reference's single-process frame loops
    (scripts/test_vel_estimation.py:145-151) turn into on a multi-GPU node; ``bench.py --gpus N`` keeps the
    one-process-per-GPU form.

    ``part_factory(device, max_frames)`` builds the per-device pipeline (default: ``FramePipeline`` on a new
    ``Context(device)``); tests inject a host-only fake to exercise the split / join logic without a GPU."""

    def __init__(self, config_manager, max_frames: int, shape: Tuple[int, int, int], devices: Optional[Sequence[int]] = None,
                 part_factory: Optional[Callable[[int, int], object]] = None, **pipeline_kwargs):
        from concurrent.futures import ThreadPoolExecutor
        if devices is None:
            devices = list(range(_lib.device_count()))
        self.devices = [int(d) for d in devices]
        if not self.devices:
            raise _lib.MmwGpuError("no HIP device visible: the MI355X HIP path is the only backend")
        self.world = len(self.devices)
        self.max_frames = int(max_frames)
        self.shape = tuple(int(x) for x in shape)
        per_dev = -(-self.max_frames // self.world)
        if part_factory is None:
            def part_factory(device, n):
                return FramePipeline(config_manager, n, shape, ctx=_lib.Context(device), **pipeline_kwargs)
        # one single-thread executor per device: every call on a device's context comes from the same host thread
        self._pools = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"mmw-dev{d}") for d in self.devices]
        self.parts = self._each(lambda r: part_factory(self.devices[r], per_dev), all_ranks=True)
        self.n_frames = 0
        self.bounds: List[Tuple[int, int]] = [(0, 0)] * self.world

    # ------------------------------------------------------------------ plumbing
    def _each(self, fn, all_ranks: bool = False) -> List:
        """``fn(rank)`` on every device thread that holds frames (or on all); results in rank order; the first
        exception is re-raised after all threads have finished."""
        ranks = [r for r in range(self.world) if all_ranks or self.bounds[r][1] > self.bounds[r][0]]
        futs = {r: self._pools[r].submit(fn, r) for r in ranks}
        out, err = [], None
        for r in ranks:
            try:
                out.append(futs[r].result())
            except Exception as e:        # noqa: BLE001 -- collected, re-raised below
                out.append(None)
                err = err or e
        if err is not None:
            raise err
        return out

    def _set_frames(self, n_frames: int):
        if n_frames > self.max_frames:
            raise ValueError("n_frames exceeds max_frames")
        self.n_frames = int(n_frames)
        self.bounds = [shard_bounds(self.n_frames, r, self.world) for r in range(self.world)]

    def _join(self, per_rank: List[List]) -> List:
        out: List = []
        for part in per_rank:
            out.extend(part)
        if len(out) != self.n_frames:
            raise RuntimeError(f"joined {len(out)} per-frame results for {self.n_frames} frames")
        return out

    def owner(self, frame: int) -> Tuple[int, int]:
        """(rank, local frame index) of a global frame index."""
        if not 0 <= frame < self.n_frames:
            raise IndexError(frame)
        r = frame * self.world // self.n_frames
        return r, frame - self.bounds[r][0]

    # ------------------------------------------------------------------ input
    def load(self, cubes: np.ndarray):
        cubes = np.asarray(cubes)
        if cubes.ndim != 4 or tuple(cubes.shape[1:]) != self.shape:
            raise ValueError(f"expected [F, {self.shape[0]}, {self.shape[1]}, {self.shape[2]}] cubes, got {cubes.shape}")
        self._set_frames(cubes.shape[0])
        self._each(lambda r: self.parts[r].load(cubes[self.bounds[r][0]:self.bounds[r][1]]))

    def synth(self, n_frames: int, seed0: int, **kw):
        """Frame f of the batch is generated from seed0 + f whichever device it lands on."""
        self._set_frames(n_frames)
        self._each(lambda r: self.parts[r].synth(self.bounds[r][1] - self.bounds[r][0], seed0 + self.bounds[r][0], **kw))

    def cubes(self) -> np.ndarray:
        V, S, C = self.shape
        out = np.empty((self.n_frames, V, S, C), dtype=np.complex64)

        def fetch(r):
            lo, hi = self.bounds[r]
            out[lo:hi] = self.parts[r].cubes(0, hi - lo)
        self._each(fetch)
        return out

    # ------------------------------------------------------------------ compute
    def detect(self) -> List[np.ndarray]:
        self.dets = self._join(self._each(lambda r: self.parts[r].detect()))
        return self.dets

    def point_clouds(self) -> List[np.ndarray]:
        pcs = self._join(self._each(lambda r: self.parts[r].point_clouds()))
        live = [p for r, p in enumerate(self.parts) if self.bounds[r][1] > self.bounds[r][0]]
        self.dets = self._join([p.dets for p in live])
        self.n_refined = sum(p.n_refined for p in live)
        return pcs

    def chain3d(self, magnitude: bool = False, out: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
        """3-D windowed FFT of every frame on its device.  ``out`` (optional, caller-owned ``[F, A, S, C]`` complex64 /
        float32 array): every device thread copies its frames into its own slice of it."""
        self._each(lambda r: self.parts[r].chain3d(magnitude))
        if out is None:
            return None
        if out.shape[0] < self.n_frames:
            raise ValueError("output array holds fewer frames than the batch")

        def fetch(r):
            lo, hi = self.bounds[r]
            for f in range(lo, hi):
                out[f] = self.parts[r].fetch_chain3d(f - lo)
        self._each(fetch)
        return out

    def fetch_chain3d(self, frame: int) -> np.ndarray:
        r, local = self.owner(frame)
        return self._pools[r].submit(self.parts[r].fetch_chain3d, local).result()

    def close(self):
        for r, pool in enumerate(self._pools):
            part = self.parts[r]

            def shut(p=part):
                if hasattr(p, "bufs"):
                    p.bufs.free()
                ctx = getattr(p, "ctx", None)
                if ctx is not None and ctx is not _lib._default_ctx:
                    ctx.close()
            try:
                pool.submit(shut).result()
            finally:
                pool.shutdown(wait=True)
        self.parts = []
