"""Ground detector: the altimeter's altitude gates the range rows (altitude ... 60-degree slant range), a 1-D Doppler CFAR
runs on |RD| of antenna 0 in every gated row
(reference: .../range_doppler_detection/range_doppler_ground_detector.py:13-127; the default ``detector_type`` of the
reference's frame-loop scripts, scripts/test_vel_estimation.py:134).

STATEFUL through its Altimeter (last-altitude gate): single-frame API only, one instance per frame sequence; it is not part
of ``batch.FramePipeline``."""
from __future__ import annotations

from typing import Dict

import numpy as np

from ...detectors.detector_registry import get_detector_registry
from ..altimeter import Altimeter
from .range_doppler_detector import RangeDopplerDetector


class RangeDopplerGroundDetector(RangeDopplerDetector):
    def __init__(self, config_manager, vel_cfar_type: str = "os_cfar_1d", vel_cfar_params: Dict = {},
                 altimeter_params: Dict = {}, **kwargs):
        super().__init__(config_manager, **kwargs)
        registry = get_detector_registry()
        if vel_cfar_type not in registry:
            raise ValueError(f"Unknown CFAR type: {vel_cfar_type}. Available: {list(registry.keys())}")
        self.vel_detector = registry[vel_cfar_type](**vel_cfar_params)
        self.altimeter_params = altimeter_params
        self.altimeter = Altimeter(config_manager, **altimeter_params)
        self.logger.info(f"RangeDopplerGroundDetector initialized with Velocity CFAR: {vel_cfar_type}")

    def reset(self):
        self.altimeter.reset()
        return super().reset()

    def _detect(self, adc_cube, rng_dop_resp, **kwargs):
        raise NotImplementedError("RangeDopplerGroundDetector uses custom process logic.")

    def process(self, adc_cube: np.ndarray, **kwargs) -> np.ndarray:
        altitude_m = self.altimeter.process(adc_cube=adc_cube, **self.altimeter_params)
        min_rng_idx = int(np.argmin(np.abs(self.range_bins - altitude_m)))
        max_rng = min(np.max(self.range_bins), altitude_m / np.cos(np.deg2rad(60)))
        max_rng_idx = int(np.argmin(np.abs(self.range_bins - max_rng)))
        rows = np.array([min_rng_idx]) if max_rng_idx == min_rng_idx else np.arange(min_rng_idx, max_rng_idx + 1)
        self._compute_range_doppler_response(adc_cube)
        self.dets = np.empty((0, 2), dtype=int)
        if len(rows) > 0:
            # the Doppler CFAR of every gated row in one launch, rows kept in order
            _, _, mask = self.vel_detector._run_rows(np.ascontiguousarray(self.rng_dop_resp[rows]))
            pairs = [(int(r), int(d)) for i, r in enumerate(rows) for d in np.where(mask[i])[0]]
            if pairs:
                self.dets = np.array(pairs, dtype=int)
        return self.dets
