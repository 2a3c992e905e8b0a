"""Range-Doppler response + detection base
(reference: mmwave_radar_processing/processors/range_doppler_detection/range_doppler_detector.py:11-116).

Per frame the device computes
  * ``rng_dop_resp_raw``  complex64 RD cube of all antennas (``mmw_range_doppler``), and
  * ``rng_dop_resp``      |RD| of virtual antenna 0 ONLY (reference :78), end-to-end in float64
                          (``mmw_range_doppler_mag64``) so the CFAR decisions match the float64 reference
                          bit for bit on detection indices.
Both stay resident in HBM for the detector / point-cloud stages; the host attributes are lazy (``_lazy.LazyAttrs``):
downloaded and converted to the reference's dtypes on first read.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from ... import _lib
from ..._lazy import LazyAttrs
from ..range_doppler_resp import RangeDopplerProcessor


class RangeDopplerDetector(LazyAttrs, RangeDopplerProcessor):
    _device_detect = False      # subclasses whose _detect can work on the device-resident planes (rng_dop_resp=None) set it

    def __init__(self, config_manager, **kwargs):
        super().__init__(config_manager)
        self.rng_dop_resp_raw: Optional[np.ndarray] = None
        self.rng_dop_resp: Optional[np.ndarray] = None
        self.dets: Optional[np.ndarray] = None
        self._dev = None        # (d_rd, d_mag64 or None, (V, S, C), d_cube) of the last frame

    def reset(self):
        super().reset()
        self.rng_dop_resp_raw = None
        self.rng_dop_resp = None
        self.dets = None
        self._dev = None

    def process(self, adc_cube: np.ndarray, **kwargs) -> np.ndarray:
        self._compute_range_doppler_device(adc_cube)
        # our own detectors decide on the device-resident planes; a user subclass written against the reference gets the
        # float64 magnitude map it expects (reference :57-59)
        self.dets = self._detect(adc_cube, None if self._device_detect else self.rng_dop_resp, **kwargs)
        return self.dets

    # ------------------------------------------------------------------ device side
    def _compute_range_doppler_device(self, adc_cube: np.ndarray):
        """Range-Doppler cube of all antennas on the device; the host copies are made when somebody reads them."""
        ctx, bufs, d_cube, d_rd, _, (V, S, C) = self._range_doppler_device(adc_cube, want_mag=False)
        self._set_device_frame(d_rd, (V, S, C), d_cube)

    def _set_device_frame(self, d_rd, shape, d_cube):
        V, S, C = shape
        self._dev = (d_rd, None, shape, d_cube)
        self._lazy_set("rng_dop_resp_raw", lambda: d_rd.download((V, S, C), np.complex64).astype(np.complex128))
        self._lazy_set("rng_dop_resp", lambda: self._mag64_device().download((S, C), np.float64))

    def _mag64_device(self):
        """|RD| of virtual antenna 0 ONLY (reference :78), end to end in float64, computed once per frame on demand."""
        d_rd, d_mag, (V, S, C), d_cube = self._dev
        if d_mag is None:
            ctx, bufs = self._device()
            d_mag = bufs.get("mag64", S * C * 8)
            _lib.check(ctx.lib.mmw_range_doppler_mag64(ctx.handle, d_cube.ptr, d_mag.ptr, 1, V, S, C, 0))
            self._dev = (d_rd, d_mag, (V, S, C), d_cube)
        return d_mag

    def _resident(self, raw) -> bool:
        """Is ``raw`` the range-Doppler cube of the frame still resident on the device (or the marker for it)?"""
        return self._dev is not None and (raw is None or raw is self.__dict__.get("rng_dop_resp_raw"))

    def _compute_range_doppler_response(self, adc_cube: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """The reference's contract (:62-80): both arrays, now (complex128 [V, S, C], float64 [S, C])."""
        self._compute_range_doppler_device(adc_cube)
        return self.rng_dop_resp_raw, self.rng_dop_resp

    def _detect(self, adc_cube: np.ndarray, rng_dop_resp: np.ndarray, **kwargs) -> np.ndarray:
        raise NotImplementedError

    def _map_detections_to_bins(self, dets: np.ndarray):
        if dets is None or dets.size == 0:
            return np.array([]), np.array([]), np.array([]), np.array([])
        r_idx = dets[:, 0].astype(int)
        v_idx = dets[:, 1].astype(int)
        return self.range_bins[r_idx], self.vel_bins[v_idx], r_idx, v_idx
