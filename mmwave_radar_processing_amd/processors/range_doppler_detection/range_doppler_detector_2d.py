"""RD response + 2-D CFAR
(reference: mmwave_radar_processing/processors/range_doppler_detection/range_doppler_detector_2d.py:12-65)."""
from __future__ import annotations

from typing import Dict

import numpy as np

from ... import _lib
from ..._lazy import LazyAttrs
from ...detectors.detector_registry import make_detector
from .range_doppler_detector import RangeDopplerDetector
from .registry import rd_detector


@rd_detector("range_doppler_detector_2d")
class RangeDopplerDetector2D(RangeDopplerDetector):
    _device_detect = True

    def __init__(self, config_manager, cfar_type: str = "ca_cfar_2d", cfar_params: Dict = {}, **kwargs):
        super().__init__(config_manager, **kwargs)
        self.detector = make_detector(cfar_type, cfar_params)
        self.points = None      # (az_idx, el_idx) of the last frame when process_points() produced them
        self.logger.info(f"RangeDopplerDetector initialized with {cfar_type} and params {cfar_params}")

    # ------------------------------------------------------------------ fused path (mmw_detect_points, one frame)
    def _cfar_args(self):
        det = self.detector
        (tr, td), (gr, gd) = det.num_train, det.num_guard
        return det.kind, int(tr), int(td), int(gr), int(gd), float(det._scale()), int(det._k_rank())

    def _fused_supported(self, S, C, n_az=0, n_el=0, num_angle_bins=64) -> bool:
        det = self.detector
        if type(self)._detect is not RangeDopplerDetector2D._detect:
            return False        # a subclass with its own _detect (the reference's extension point) is never bypassed
        if not hasattr(det, "_launch_device") or not hasattr(det, "num_train_cells"):
            return False
        kind, tr, td, gr, gd, _, _ = self._cfar_args()
        ctx, _ = self._device()
        return bool(ctx.lib.mmw_detect_points_supported(S, C, kind, tr, td, gr, gd, n_az, n_el, int(num_angle_bins)))

    def process_points(self, adc_cube: np.ndarray, az=(), el=(), shift_az=True, shift_el=False, num_angle_bins=64):
        """Detections AND the argmax angle bins of both antenna lists in one device call (``mmw_detect_points``); returns
        ``dets`` and leaves ``self.points = (az_idx, el_idx)`` (None for an empty list).  None when the request has no
        fused kernel (the caller then takes ``process()`` + ``mmw_angle_argmax_exact``)."""
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        if not self._fused_supported(S, C, len(az), len(el), num_angle_bins):
            return None
        L, h = ctx.lib, ctx.handle
        cap = S * C
        d_rd, d_l1 = bufs.get("rd", V * S * C * 8), bufs.get("plane_l1", V * 4)
        d_dets, d_cnt = bufs.get("dets", cap * 8), bufs.get("count", 4)
        d_az = bufs.get("az_idx", cap * 4) if len(az) else None
        d_el = bufs.get("el_idx", cap * 4) if len(el) else None
        kind, tr, td, gr, gd, scale, k_rank = self._cfar_args()
        a_az, n_az = _lib.int_array(az)
        a_el, n_el = _lib.int_array(el)
        stats = (_lib.C.c_int * 5)()
        _lib.check(L.mmw_detect_points(h, d_cube.ptr, d_rd.ptr, d_l1.ptr, None, d_dets.ptr, d_cnt.ptr,
                                       d_az.ptr if d_az else None, d_el.ptr if d_el else None, 1, V, S, C, kind, tr, td, gr, gd,
                                       scale, k_rank, cap, a_az, n_az, int(bool(shift_az)), a_el, n_el, int(bool(shift_el)),
                                       int(num_angle_bins), stats))
        self.screen_stats = list(stats)     # frames / cells undecided by the screening, handed back, az / el refined in float64
        count = int(d_cnt.download((1,), np.int32)[0])
        if count < 0:
            return None         # the screening pass handed the frame back (non-finite samples ...): float64 path
        self._set_device_frame(d_rd, (V, S, C), d_cube)
        self._dev_dets = (d_dets, d_cnt, cap, count)
        self._arm_cfar_thunks(S, C)
        self.dets = d_dets.download((count, 2), np.int32).astype(int) if count else np.empty((0, 2), dtype=int)
        self.points = (d_az.download((count,), np.int32).astype(int) if d_az and count else (np.empty(0, int) if d_az else None),
                       d_el.download((count,), np.int32).astype(int) if d_el and count else (np.empty(0, int) if d_el else None))
        return self.dets

    def _arm_cfar_thunks(self, S, C):
        """thresholds / noise estimates / boolean map of the CFAR object: one float64 kernel run on first read of any."""
        det = self.detector
        if not isinstance(det, LazyAttrs):
            return
        state = {}

        def run():
            if not state:
                ctx, bufs = self._device()
                n = S * C
                d_thr, d_noise, d_mask = bufs.get("thr", n * 8), bufs.get("noise", n * 8), bufs.get("mask", n)
                det._launch_device(ctx, self._mag64_device().ptr, d_thr.ptr, d_noise.ptr, d_mask.ptr, 1, S, C)
                state["thr"] = d_thr.download((S, C), np.float64)
                state["noise"] = d_noise.download((S, C), np.float64)
                state["det"] = d_mask.download((S, C), np.uint8).astype(bool)
            return state

        det._lazy_set("thresholds", lambda: run()["thr"])
        det._lazy_set("noise_estimates", lambda: run()["noise"])
        det._lazy_set("detections", lambda: run()["det"])

    def reset(self):
        super().reset()
        self.points = None
        det = self.detector
        if isinstance(det, LazyAttrs):      # the thunks point at the frame that has just been dropped
            det._lazy_clear("thresholds", "noise_estimates", "detections")
            det.thresholds = det.noise_estimates = det.detections = None

    def process(self, adc_cube: np.ndarray, **kwargs) -> np.ndarray:
        self.points = None
        dets = self.process_points(adc_cube)
        if dets is not None:
            return dets
        return super().process(adc_cube, **kwargs)

    def _detect(self, adc_cube: np.ndarray, rng_dop_resp: np.ndarray, **kwargs) -> np.ndarray:
        """CFAR on the device-resident float64 plane, ordered compaction, -> int64 (N, 2) [range_idx, doppler_idx]."""
        det = self.detector
        on_device = self._dev is not None and hasattr(det, "_launch_device") and \
            (rng_dop_resp is None or rng_dop_resp is self.__dict__.get("rng_dop_resp"))
        if not on_device:
            if rng_dop_resp is None:
                rng_dop_resp = self.rng_dop_resp
            dets = det.detect(rng_dop_resp)     # foreign map or 1-D detector: the detector's own path
            return np.array(dets, dtype=int) if dets else np.empty((0, 2), dtype=int)
        ctx, bufs = self._device()
        (_, S, C) = self._dev[2]
        n = S * C
        d_mask = bufs.get("mask", n)
        det._launch_device(ctx, self._mag64_device().ptr, None, None, d_mask.ptr, 1, S, C)      # decision only
        cap = n
        d_dets, d_cnt = bufs.get("dets", cap * 8), bufs.get("count", 4)
        _lib.check(ctx.lib.mmw_compact2d(ctx.handle, d_mask.ptr, d_dets.ptr, d_cnt.ptr, 1, S, C, cap))
        count = int(d_cnt.download((1,), np.int32)[0])
        self._dev_dets = (d_dets, d_cnt, cap, count)
        self._arm_cfar_thunks(S, C)
        if count == 0:
            return np.empty((0, 2), dtype=int)
        return d_dets.download((count, 2), np.int32).astype(int)
