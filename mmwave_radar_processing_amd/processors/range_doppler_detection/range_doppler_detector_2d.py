"""RD response + 2-D CFAR
(reference: mmwave_radar_processing/processors/range_doppler_detection/range_doppler_detector_2d.py:12-65)."""
from __future__ import annotations

from typing import Dict

import numpy as np

from ... import _lib
from ...detectors.detector_registry import get_detector_registry
from .range_doppler_detector import RangeDopplerDetector


class RangeDopplerDetector2D(RangeDopplerDetector):
    def __init__(self, config_manager, cfar_type: str = "ca_cfar_2d", cfar_params: Dict = {}, **kwargs):
        super().__init__(config_manager, **kwargs)
        registry = get_detector_registry()
        if cfar_type not in registry:
            raise ValueError(f"Unknown CFAR type: {cfar_type}. Available: {list(registry.keys())}")
        self.detector = registry[cfar_type](**cfar_params)
        self.logger.info(f"RangeDopplerDetector initialized with {cfar_type} and params {cfar_params}")

    def _detect(self, adc_cube: np.ndarray, rng_dop_resp: np.ndarray, **kwargs) -> np.ndarray:
        """CFAR on the device-resident float64 plane, ordered compaction, -> int64 (N, 2) [range_idx, doppler_idx]."""
        det = self.detector
        if self._dev is None or not hasattr(det, "_launch_device") or rng_dop_resp is not self.rng_dop_resp:
            dets = det.detect(rng_dop_resp)     # foreign map or 1-D detector: the detector's own path
            return np.array(dets, dtype=int) if dets else np.empty((0, 2), dtype=int)
        ctx, bufs = self._device()
        _, d_mag, (_, S, C) = self._dev[:3]
        n = S * C
        d_thr, d_noise, d_mask = bufs.get("thr", n * 8), bufs.get("noise", n * 8), bufs.get("mask", n)
        det._launch_device(ctx, d_mag.ptr, d_thr.ptr, d_noise.ptr, d_mask.ptr, 1, S, C)
        cap = n
        d_dets, d_cnt = bufs.get("dets", cap * 8), bufs.get("count", 4)
        _lib.check(ctx.lib.mmw_compact2d(ctx.handle, d_mask.ptr, d_dets.ptr, d_cnt.ptr, 1, S, C, cap))
        count = int(d_cnt.download((1,), np.int32)[0])
        self._dev_dets = (d_dets, d_cnt, cap, count)
        det.thresholds = d_thr.download((S, C), np.float64)
        det.noise_estimates = d_noise.download((S, C), np.float64)
        det.detections = d_mask.download((S, C), np.uint8).astype(bool)
        if count == 0:
            return np.empty((0, 2), dtype=int)
        return d_dets.download((count, 2), np.int32).astype(int)
