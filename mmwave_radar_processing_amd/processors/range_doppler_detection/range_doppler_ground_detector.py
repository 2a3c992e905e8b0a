"""Ground detector: the altimeter's altitude gates the range rows (altitude ... 60-degree slant range), a 1-D Doppler CFAR
runs on |RD| of antenna 0 in every gated row
(reference: .../range_doppler_detection/range_doppler_ground_detector.py:13-127; the default ``detector_type`` of the
reference's frame-loop scripts, scripts/test_vel_estimation.py:134).

STATEFUL through its Altimeter (last-altitude gate): single-frame API only, one instance per frame sequence; it is not part
of ``batch.FramePipeline``."""
from __future__ import annotations

from typing import Dict

import numpy as np

from ...detectors.base import BaseCFAR1D
from ...detectors.detector_registry import make_detector
from ..altimeter import Altimeter
from .range_doppler_detector import RangeDopplerDetector
from .registry import rd_detector


def slant_gate(range_bins: np.ndarray, altitude_m: float, max_off_nadir_deg: float = 60.0) -> np.ndarray:
    """Range rows the ground can occupy: from the bin nearest the altitude out to the bin nearest the slant range at
    ``max_off_nadir_deg`` (capped at the last bin), both ends included."""
    far_m = min(range_bins[-1], altitude_m / np.cos(np.deg2rad(max_off_nadir_deg)))
    near, far = (int(np.abs(range_bins - m).argmin()) for m in (altitude_m, far_m))
    return np.arange(near, far + 1)


@rd_detector("range_doppler_ground_detector")       # stateful (Altimeter): single-frame API only
class RangeDopplerGroundDetector(RangeDopplerDetector):
    def __init__(self, config_manager, vel_cfar_type: str = "os_cfar_1d", vel_cfar_params: Dict = {},
                 altimeter_params: Dict = {}, **kwargs):
        super().__init__(config_manager, **kwargs)
        self.vel_detector = make_detector(vel_cfar_type, vel_cfar_params)
        self.altimeter_params = altimeter_params
        self.altimeter = Altimeter(config_manager, **altimeter_params)
        self.logger.info(f"RangeDopplerGroundDetector initialized with Velocity CFAR: {vel_cfar_type}")

    def reset(self):
        self.altimeter.reset()
        return super().reset()

    def _detect(self, adc_cube, rng_dop_resp, **kwargs):
        raise NotImplementedError("RangeDopplerGroundDetector uses custom process logic.")

    def process(self, adc_cube: np.ndarray, **kwargs) -> np.ndarray:
        rows = slant_gate(self.range_bins, self.altimeter.process(adc_cube=adc_cube, **self.altimeter_params))
        self._compute_range_doppler_response(adc_cube)
        pairs = self._doppler_cfar(rows, np.ascontiguousarray(self.rng_dop_resp[rows]))
        self.dets = np.array(pairs, dtype=int).reshape(-1, 2)
        return self.dets

    def _doppler_cfar(self, rows, mag_rows):
        """(range row, Doppler bin) pairs of the gated rows, rows in order.  The stock 1-D detectors take all rows in ONE
        launch and are left in the state the reference's per-row loop leaves them in (the last row's thresholds / noise /
        decisions); a detector with its own ``_compute_thresholds`` (the subclass hook) or without the row kernel -- e.g. a
        2-D registry key, which raises the reference's ValueError -- is called row by row through ``detect``."""
        det = self.vel_detector
        if len(rows) == 0:
            return []
        if getattr(type(det), "_compute_thresholds", None) is not BaseCFAR1D._compute_thresholds:
            return [(int(r), int(d)) for r, row in zip(rows, mag_rows) for d in det.detect(row)]
        thr, noise, mask = det._run_rows(mag_rows)
        det.thresholds, det.noise_estimates, det.detections = thr[-1], noise[-1], mask[-1]
        return [(int(r), int(d)) for i, r in enumerate(rows) for d in np.where(mask[i])[0]]
