"""Coarse range profile (reference: mmwave_radar_processing/processors/range_resp.py:8-57,153-164).

The FFT and the ZoomFFT (:59-102, as a chirp-z transform) run on the device; the scipy peak picking (:104-149) that
``Altimeter`` builds on stays on the host, as in the reference.
"""
from __future__ import annotations

import numpy as np
from scipy.signal import find_peaks as _scipy_find_peaks

from .. import _lib
from ._processor import _Processor


class RangeProcessor(_Processor):
    def __init__(self, config_manager, **kwargs):
        self.num_range_bins = None
        self.range_bins = None
        super().__init__(config_manager)

    def configure(self):
        cm = self.config_manager
        self.num_range_bins = cm.get_num_adc_samples(profile_idx=0)
        self.range_bins = np.arange(start=0, step=cm.range_res_m, stop=cm.range_max_m - cm.range_res_m / 2)

    def coarse_fft(self, adc_cube: np.ndarray, chirp_idx: int = 0) -> np.ndarray:
        """mean over rx of |FFT_S(hann(S) * x[:, :, chirp_idx])| -> float64 (S,)."""
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        d_out = bufs.get("profile", S * 4)
        _lib.check(ctx.lib.mmw_range_profile(ctx.handle, d_cube.ptr, d_out.ptr, 1, V, S, C, int(chirp_idx)))
        return d_out.download((S,), np.float32).astype(np.float64)

    def zoom_fft(self, adc_cube: np.ndarray, range_start_m: float, range_stop_m: float, chirp_idx: int = 0):
        """(zoom_fft_magnitude, zoom_range_bins): S bins between the two ranges, mean |.| over antennas (:59-102)."""
        cm = self.config_manager
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        fs = 1 / cm.range_res_m
        freq_start = range_start_m * fs / cm.range_max_m
        freq_stop = range_stop_m * fs / cm.range_max_m
        m = S                                              # ZoomFFT(n, fn, m=None) -> m = n, endpoint=False
        f0 = freq_start / fs
        df = (freq_stop - freq_start) / m / fs
        d_out = bufs.get("zoom", m * 4)
        _lib.check(ctx.lib.mmw_range_zoom(ctx.handle, d_cube.ptr, d_out.ptr, 1, V, S, C, int(chirp_idx), m,
                                          float(f0), float(df)))
        zoom_avg = d_out.download((m,), np.float32).astype(np.float64)
        return zoom_avg, np.linspace(range_start_m, range_stop_m, m)

    def find_peaks(self, rng_resp_db: np.ndarray, rng_bins: np.ndarray, max_peaks: int = 3, threshold_dB: int = 20):
        """Host-side peak picker on a range profile in dB (reference :104-149; ``Altimeter(RangeProcessor)`` relies on
        it): local maxima of >= 6 dB prominence, within ``threshold_dB`` of the strongest, strongest first, at most
        ``max_peaks``.  Returns (ranges in metres, dB values); two empty arrays when there is no peak."""
        idx, _ = _scipy_find_peaks(rng_resp_db, prominence=6)
        if len(idx) == 0:
            return np.array([]), np.array([])
        vals = rng_resp_db[idx]
        keep = vals >= (np.max(vals) - threshold_dB)
        idx, vals = idx[keep], vals[keep]
        top = idx[np.argsort(vals)[::-1]][:max_peaks]
        return rng_bins[top], rng_resp_db[top]

    def process(self, adc_cube: np.ndarray, chirp_idx: int = 0, **kwargs) -> np.ndarray:
        return self.coarse_fft(adc_cube, chirp_idx)
