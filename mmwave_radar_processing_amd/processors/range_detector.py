"""Range profile + 1-D CFAR (reference: mmwave_radar_processing/processors/range_detector.py:11-97)."""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _lib
from ..detectors.detector_registry import get_detector_registry
from .range_resp import RangeProcessor


class RangeDetector(RangeProcessor):
    def __init__(self, config_manager, cfar_type: str = "os_cfar_1d", cfar_params: dict = {}, **kwargs):
        registry = get_detector_registry()
        if cfar_type not in registry:
            raise ValueError(f"Unknown CFAR type: {cfar_type}. Available: {list(registry.keys())}")
        self.cfar_detector = registry[cfar_type](**cfar_params)
        self.dets: Optional[np.ndarray] = None
        self.thresholds: Optional[np.ndarray] = None
        self.range_resp: Optional[np.ndarray] = None
        super().__init__(config_manager)
        self.logger.info(f"RangeDetector initialized with CFAR type: {cfar_type}")

    def coarse_fft(self, adc_cube: np.ndarray, chirp_idx: int = 0) -> np.ndarray:
        """The profile the CFAR thresholds is float64 end to end (``mmw_range_profile_f64``): a float32 profile would move
        borderline cells across the reference's float64 threshold."""
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        d_out = bufs.get("profile64", S * 8)
        _lib.check(ctx.lib.mmw_range_profile_f64(ctx.handle, d_cube.ptr, d_out.ptr, 1, V, S, C, int(chirp_idx)))
        return d_out.download((S,), np.float64)

    def process(self, adc_cube: np.ndarray, **kwargs) -> np.ndarray:
        self.range_resp = super().process(adc_cube, chirp_idx=0)
        det_range_idxs = self.cfar_detector.detect(self.range_resp)
        self.thresholds = self.cfar_detector.thresholds
        self.dets = np.array(det_range_idxs) if len(det_range_idxs) > 0 else np.array([])
        return self.dets

    def _map_detections_to_bins(self, dets: np.ndarray) -> np.ndarray:
        """Range of every detection index.  (The reference's method stops after the configuration check, :85-97, and
        returns None; this one finishes the lookup its docstring describes.)"""
        if self.range_bins is None:
            raise ValueError("Range bins are not configured.")
        dets = np.asarray(dets)
        if dets.size == 0:
            return np.array([])
        return self.range_bins[dets.astype(int)]
