"""Range to ground from the chirp-0 range profile, tracked from frame to frame.

Behaviour pinned on sequences run by the reference (mmwave_radar_processing/processors/altimeter.py, fixtures
``tests/golden/detectors_rd.npz: ground_*``; a restatement of the reference's control flow exists only in the test
infrastructure).  Here the tracker is a small state machine:

* ``GroundLock`` holds the gate (lowest admissible range, how far a new measurement may lie from the last one) and the
  two altitudes the callers read;
* a frame is a list of *look stages* -- coarse profile, then optionally a zoom around the coarse hit -- each turning the
  cube into candidate ranges on the device (float64 range profile / chirp-z zoom) + one scipy peak pick on the host;
* every stage's candidates pass through ``GroundLock.admit``; the first stage with no admissible candidate ends the frame
  with the previous altitude, the last stage's hit is accepted.

STATEFUL: one instance per frame sequence, not part of the batch API.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import numpy as np

from .. import _lib
from .range_resp import RangeProcessor


@dataclass
class GroundLock:
    floor_m: float                      # nothing nearer than this is ground
    reach_m: float                      # a new measurement lies within this of the last one
    bias_m: float = 0.0                 # added to what is reported, never to the gate
    measured_m: float = field(init=False)
    reported_m: float = field(init=False)

    def __post_init__(self):
        self.measured_m = self.reported_m = self.floor_m

    def admit(self, candidates_m: np.ndarray) -> Optional[float]:
        """The nearest candidate inside the gate, or None."""
        c = np.asarray(candidates_m, dtype=float)
        inside = c[(c >= self.floor_m) & (np.abs(c - self.measured_m) <= self.reach_m)]
        return float(inside.min()) if inside.size else None

    def accept(self, range_m: float) -> None:
        self.measured_m = range_m
        self.reported_m = range_m + self.bias_m

    def release(self) -> None:
        """Forget the last measurement (the gate re-centres on the floor); the reported altitude stands until the next
        accepted frame, which is what the reference's reset leaves behind."""
        self.measured_m = self.floor_m


class Altimeter(RangeProcessor):
    def __init__(self, config_manager, min_altitude_m: float, zoom_search_region_m: float, altitude_search_limit_m: float,
                 range_bias: float = 0.0, **kwargs) -> None:
        super().__init__(config_manager)
        self.lock = GroundLock(float(min_altitude_m), float(altitude_search_limit_m), float(range_bias))
        self.zoom_half_width_m = float(zoom_search_region_m)
        self.coarse_fft_data = None

    # the names callers of the reference class read ---------------------------------------------------------------
    current_altitude_measured_m = property(lambda self: self.lock.measured_m)
    current_altitude_corrected_m = property(lambda self: self.lock.reported_m)
    min_altitude_m = property(lambda self: self.lock.floor_m)
    altitude_search_limit_m = property(lambda self: self.lock.reach_m)
    range_bias = property(lambda self: self.lock.bias_m)
    zoom_search_region_m = property(lambda self: self.zoom_half_width_m)

    def find_ground_peak(self, detected_peaks_m: np.ndarray) -> float:
        hit = self.lock.admit(detected_peaks_m)
        return -1.0 if hit is None else hit

    def reset(self):
        self.lock.release()
        return super().reset()

    # look stages: (cube, previous hit) -> candidate ranges ---------------------------------------------------------
    def _look_coarse(self, adc_cube: np.ndarray, _prev) -> np.ndarray:
        # float64 on the device: the 6-dB prominence test then sees the reference's numbers to ~1e-15
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        d_prof = bufs.get("profile64", S * 8)
        _lib.check(ctx.lib.mmw_range_profile_f64(ctx.handle, d_cube.ptr, d_prof.ptr, 1, V, S, C, 0))
        self.coarse_fft_data = d_prof.download((S,), np.float64)
        return self.find_peaks(20 * np.log10(self.coarse_fft_data), self.range_bins, max_peaks=3)[0]

    def _look_zoom(self, adc_cube: np.ndarray, prev_m: float) -> np.ndarray:
        lo = max(1e-6, prev_m - self.zoom_half_width_m)
        hi = min(self.range_bins.max() - 1e-6, prev_m + self.zoom_half_width_m)
        spectrum, bins = self.zoom_fft(adc_cube, lo, hi, 0)
        return self.find_peaks(20 * np.log10(spectrum), bins, max_peaks=2)[0]

    def process(self, adc_cube: np.ndarray, precise_est_enabled: bool = True, **kwargs) -> float:
        stages: List[Callable] = [self._look_coarse] + ([self._look_zoom] if precise_est_enabled else [])
        hit = None
        for look in stages:
            hit = self.lock.admit(look(adc_cube, hit))
            if hit is None:
                return self.lock.reported_m
        self.lock.accept(hit)
        return self.lock.reported_m
