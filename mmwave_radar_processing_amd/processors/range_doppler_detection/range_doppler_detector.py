"""Range-Doppler response + detection base
(reference: mmwave_radar_processing/processors/range_doppler_detection/range_doppler_detector.py:11-116).

Per frame the device computes
  * ``rng_dop_resp_raw``  complex64 RD cube of all antennas (``mmw_range_doppler``), and
  * ``rng_dop_resp``      |RD| of virtual antenna 0 ONLY (reference :78), end-to-end in float64
                          (``mmw_range_doppler_mag64``) so the CFAR decisions match the float64 reference
                          bit for bit on detection indices.
Both stay resident in HBM for the detector / point-cloud stages; host copies are made on demand.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from ... import _lib
from ..range_doppler_resp import RangeDopplerProcessor


class RangeDopplerDetector(RangeDopplerProcessor):
    def __init__(self, config_manager, **kwargs):
        super().__init__(config_manager)
        self.rng_dop_resp_raw: Optional[np.ndarray] = None
        self.rng_dop_resp: Optional[np.ndarray] = None
        self.dets: Optional[np.ndarray] = None
        self._dev = None        # (d_rd, d_mag64, (V, S, C), d_cube) of the last frame

    def reset(self):
        super().reset()
        self.rng_dop_resp_raw = None
        self.rng_dop_resp = None
        self.dets = None
        self._dev = None

    def process(self, adc_cube: np.ndarray, **kwargs) -> np.ndarray:
        self._compute_range_doppler_response(adc_cube)
        self.dets = self._detect(adc_cube, self.rng_dop_resp, **kwargs)
        return self.dets

    def _compute_range_doppler_response(self, adc_cube: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        ctx, bufs, d_cube, d_rd, _, (V, S, C) = self._range_doppler_device(adc_cube, want_mag=False)
        d_mag = bufs.get("mag64", S * C * 8)
        _lib.check(ctx.lib.mmw_range_doppler_mag64(ctx.handle, d_cube.ptr, d_mag.ptr, 1, V, S, C, 0))
        self._dev = (d_rd, d_mag, (V, S, C), d_cube)
        self.rng_dop_resp_raw = d_rd.download((V, S, C), np.complex64).astype(np.complex128)
        self.rng_dop_resp = d_mag.download((S, C), np.float64)
        return self.rng_dop_resp_raw, self.rng_dop_resp

    def _detect(self, adc_cube: np.ndarray, rng_dop_resp: np.ndarray, **kwargs) -> np.ndarray:
        raise NotImplementedError

    def _map_detections_to_bins(self, dets: np.ndarray):
        if dets is None or dets.size == 0:
            return np.array([]), np.array([]), np.array([]), np.array([])
        r_idx = dets[:, 0].astype(int)
        v_idx = dets[:, 1].astype(int)
        return self.range_bins[r_idx], self.vel_bins[v_idx], r_idx, v_idx
