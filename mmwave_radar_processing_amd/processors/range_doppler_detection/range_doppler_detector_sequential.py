"""Sequential detector: 1-D range CFAR on the chirp-0 range profile, then 1-D Doppler CFAR on each detected
range row (reference: .../range_doppler_detection/range_doppler_detector_sequential.py:12-107)."""
from __future__ import annotations

from typing import Dict

import numpy as np

from ... import _lib
from ...detectors.detector_registry import make_detector
from ..range_resp import RangeProcessor
from .range_doppler_detector import RangeDopplerDetector
from .registry import rd_detector


@rd_detector("range_doppler_detector_sequential")
class RangeDopplerDetectorSequential(RangeDopplerDetector):
    def __init__(self, config_manager, rng_cfar_type: str = "os_cfar_1d", rng_cfar_params: Dict = {},
                 vel_cfar_type: str = "os_cfar_1d", vel_cfar_params: Dict = {}, **kwargs):
        super().__init__(config_manager, **kwargs)
        self.rng_detector = make_detector(rng_cfar_type, rng_cfar_params)
        self.vel_detector = make_detector(vel_cfar_type, vel_cfar_params)
        self.range_processor = RangeProcessor(config_manager)
        self.logger.info(f"RangeDopplerDetectorSequential initialized with Range CFAR: {rng_cfar_type}, "
                         f"Velocity CFAR: {vel_cfar_type}")

    def _detect(self, adc_cube, rng_dop_resp, **kwargs):
        raise NotImplementedError("RangeDopplerDetectorSequential uses custom process logic.")

    def process(self, adc_cube: np.ndarray, **kwargs) -> np.ndarray:
        # float64 range profile so the range CFAR decides exactly like the float64 reference
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        d_prof = bufs.get("profile64", S * 8)
        _lib.check(ctx.lib.mmw_range_profile_f64(ctx.handle, d_cube.ptr, d_prof.ptr, 1, V, S, C, 0))
        range_resp = d_prof.download((S,), np.float64)
        det_range_idxs = self.rng_detector.detect(x=range_resp)
        self._compute_range_doppler_response(adc_cube)
        self.dets = np.empty((0, 2), dtype=int)
        if len(det_range_idxs) > 0:
            # all Doppler rows in one launch, then keep the detected range rows in order
            _, _, mask = self.vel_detector._run_rows(self.rng_dop_resp)
            pairs = [(r, int(d)) for r in det_range_idxs for d in np.where(mask[r])[0]]
            if pairs:
                self.dets = np.array(pairs, dtype=int)
        return self.dets
