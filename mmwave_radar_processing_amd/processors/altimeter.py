"""Altitude (range to ground) from the chirp-0 range profile
(reference: mmwave_radar_processing/processors/altimeter.py:6-140).

STATEFUL: the last measured altitude gates the next frame's ground peak (:42-65), so an instance belongs to ONE frame
sequence and stays out of the batch API.  The transforms run on the device (float64 range profile,
chirp-z zoom transform); the peak picking is the reference's scipy call on the host.
"""
from __future__ import annotations

import numpy as np

from .. import _lib
from .range_resp import RangeProcessor


class Altimeter(RangeProcessor):
    def __init__(self, config_manager, min_altitude_m: float, zoom_search_region_m: float, altitude_search_limit_m: float,
                 range_bias: float = 0.0, **kwargs) -> None:
        super().__init__(config_manager)
        self.min_altitude_m = float(min_altitude_m)
        self.zoom_search_region_m = float(zoom_search_region_m)
        self.altitude_search_limit_m = float(altitude_search_limit_m)
        self.range_bias = float(range_bias)
        self.coarse_fft_data = None
        self.current_altitude_measured_m = self.min_altitude_m      # measured by the radar
        self.current_altitude_corrected_m = self.min_altitude_m     # corrected for the bias

    def reset(self):
        self.current_altitude_measured_m = self.min_altitude_m      # (the corrected value is kept, as in the reference :37-40)
        return super().reset()

    def find_ground_peak(self, detected_peaks_m: np.ndarray):
        """Nearest valid peak: at or above the minimum altitude and within the search limit of the current one; -1.0 when
        there is none (reference :42-65)."""
        if detected_peaks_m.size > 0:
            valid = detected_peaks_m[(detected_peaks_m >= self.min_altitude_m) &
                                     (np.abs(detected_peaks_m - self.current_altitude_measured_m) <= self.altitude_search_limit_m)]
            if valid.size > 0:
                return np.min(valid)
        return -1.0

    def _perform_coarse_fft(self, adc_cube: np.ndarray) -> np.ndarray:
        # float64 on the device: the 6-dB prominence test of find_peaks then sees the reference's numbers to ~1e-15
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        d_out = bufs.get("profile64", S * 8)
        _lib.check(ctx.lib.mmw_range_profile_f64(ctx.handle, d_cube.ptr, d_out.ptr, 1, V, S, C, 0))
        return d_out.download((S,), np.float64)

    def _get_coarse_peaks(self, coarse_fft: np.ndarray) -> np.ndarray:
        peaks, _ = self.find_peaks(rng_resp_db=20 * np.log10(coarse_fft), rng_bins=self.range_bins, max_peaks=3)
        return peaks

    def _refine_altitude_estimate(self, adc_cube: np.ndarray, ground_peak: float) -> float:
        range_start_m = max(1e-6, ground_peak - self.zoom_search_region_m)
        range_end_m = min(np.max(self.range_bins) - 1e-6, ground_peak + self.zoom_search_region_m)
        zoom_avg, zoom_bins = self.zoom_fft(adc_cube=adc_cube, range_start_m=range_start_m, range_stop_m=range_end_m,
                                            chirp_idx=0)
        peaks, _ = self.find_peaks(rng_resp_db=20 * np.log10(zoom_avg), rng_bins=zoom_bins, max_peaks=2)
        return self.find_ground_peak(detected_peaks_m=peaks) if peaks.size > 0 else -1.0

    def process(self, adc_cube: np.ndarray, precise_est_enabled: bool = True, **kwargs):
        self.coarse_fft_data = self._perform_coarse_fft(adc_cube)
        peaks = self._get_coarse_peaks(self.coarse_fft_data)
        if peaks.size == 0:
            return self.current_altitude_corrected_m
        ground_peak = self.find_ground_peak(detected_peaks_m=peaks)
        if ground_peak < 0:
            return self.current_altitude_corrected_m
        if not precise_est_enabled:
            self.current_altitude_measured_m = ground_peak
            self.current_altitude_corrected_m = ground_peak + self.range_bias
            return self.current_altitude_corrected_m
        refined = self._refine_altitude_estimate(adc_cube, ground_peak)
        if refined > 0:
            self.current_altitude_measured_m = refined
            self.current_altitude_corrected_m = refined + self.range_bias
        return self.current_altitude_corrected_m
