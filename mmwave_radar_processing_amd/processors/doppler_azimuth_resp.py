"""Doppler-azimuth response, coarse FFT and precise (ZoomFFT) modes
(reference: mmwave_radar_processing/processors/doppler_azimuth_resp.py:8-334,419-491).

antenna subset -> Hann(S) x Hann(C) (x Hann(V) on the "standard" geometry with virtual antennas) -> range FFT ->
keep the range bins inside ``range_window`` -> 2-D FFT over (chirp, zero-padded antenna) -> fftshift -> |.| ->
valid angle columns -> mean over the kept range bins.  On the GPU this is ``mmw_doppler_azimuth``: the range-Doppler
kernel followed by one kernel that does the angle FFT, |.| and the range mean without writing the magnitude cube.

``use_precise_fft=True`` replaces the Doppler FFT by the reference's two ``scipy.signal.ZoomFFT`` calls (one per
velocity sign) over ``precise_vel_range``; the host derives the frequency of every zoomed bin exactly as the reference
does and ``mmw_doppler_azimuth_zoom`` evaluates range FFT -> zoom transform -> angle FFT -> |.| -> range mean on the
GPU.  ``detect_peaks_rows`` / ``detect_peak_zero_az`` (:336-417) are scipy peak pickers over the small
``[vel, angle]`` map this processor returns; they run on the host exactly as in the reference and exist so that
subclasses written against the reference (``VelocityEstimator(DopplerAzimuthProcessor)``, velocity_estimator.py:7)
keep working when the base class is swapped.
"""
from __future__ import annotations

from typing import Union

import numpy as np
from scipy.signal import find_peaks

from .. import _lib
from ._processor import _Processor
from .range_angle_resp import angle_tables


class DopplerAzimuthProcessor(_Processor):
    def __init__(self, config_manager, num_angle_bins: int = 64,
                 valid_angle_range: np.ndarray = np.array([np.deg2rad(-60), np.deg2rad(60)]),
                 min_zoom_fft_vel_span=0.1, **kwargs) -> None:
        self.num_range_bins = None
        self.range_bins = None
        self.vel_bins = None
        self.zoomed_vel_bins = None
        self.min_zoom_fft_vel_span = min_zoom_fft_vel_span
        self.num_angle_bins = num_angle_bins
        self.phase_shifts = None
        self.angle_bins = None
        self.valid_angle_range = np.asarray(valid_angle_range, dtype=float)
        self.valid_angle_mask = None
        self.valid_angle_bins = None
        super().__init__(config_manager)

    def configure(self):
        cm = self.config_manager
        self.num_range_bins = cm.get_num_adc_samples(profile_idx=0)
        self.range_bins = np.arange(start=0, step=cm.range_res_m, stop=cm.range_max_m - cm.range_res_m / 2 + 1e-3)
        self.vel_bins = np.arange(start=-1 * cm.vel_max_m_s, stop=cm.vel_max_m_s - cm.vel_res_m_s + 1e-3,
                                  step=cm.vel_res_m_s)
        self.num_rx_antennas = cm.num_rx_antennas
        self.phase_shifts, self.angle_bins = angle_tables(self.num_angle_bins)
        self.valid_angle_mask = (self.angle_bins >= self.valid_angle_range[0]) & \
            (self.angle_bins <= self.valid_angle_range[1])
        self.valid_angle_bins = self.angle_bins[self.valid_angle_mask]

    def set_zoomed_fft_vel_bins(self, vel_range) -> np.ndarray:
        """Velocity bins of the precise mode: ``num_vel_bins`` below zero and ``num_vel_bins`` above (:165-203)."""
        n = self.vel_bins.size
        neg = np.linspace(start=vel_range[0], stop=min(-1e-4, vel_range[1]), num=n if vel_range[0] <= 0 else 0,
                          endpoint=False)
        pos = np.linspace(start=max(1e-4, vel_range[0]), stop=vel_range[1], num=n if vel_range[1] > 0 else 0,
                          endpoint=False)
        self.zoomed_vel_bins = np.concatenate((neg, pos))
        return self.zoomed_vel_bins

    def _zoom_plan(self, vel_range) -> np.ndarray:
        """Clamp / widen the velocity range (:234-246), set ``zoomed_vel_bins`` and return, per zoomed bin, the
        frequency in cycles per chirp of the reference's ZoomFFT call for its half (:148-155,254-283); NaN where the
        reference emits zeros because the half spans less than ``min_zoom_fft_vel_span``.

        The caller's array is left untouched (the reference edits ``precise_vel_range`` in place)."""
        cm = self.config_manager
        vmax = cm.vel_max_m_s
        vr = np.array(vel_range, dtype=float).ravel()
        vr[0] = max(vr[0], -1 * vmax)
        vr[1] = min(vr[1], vmax)
        spread = 2 * self.min_zoom_fft_vel_span
        if (vr[1] - vr[0]) < spread:
            to_max, to_min = abs(vr[1] - vmax), abs(vr[0] + vmax)
            if to_max > to_min:
                vr[1] = vr[0] + spread
            elif to_min > to_max:
                vr[0] = vr[1] - spread
        bins = self.set_zoomed_fft_vel_bins(vr)
        fs = 1 / cm.vel_res_m_s
        freq = np.full(bins.size, np.nan)
        n_neg = int(np.count_nonzero(bins <= 0))
        for vals, first, alias in ((bins[:n_neg], 0, 2 * vmax), (bins[n_neg:], n_neg, 0.0)):
            m = vals.size
            if m > 0 and np.abs(np.max(vals) - np.min(vals)) > self.min_zoom_fft_vel_span:
                f1 = (np.min(vals) + alias) * fs / vmax
                f2 = (np.max(vals) + alias) * fs / vmax
                fz = fs * 2                       # the reference's "factor of 2" (:154-155)
                freq[first:first + m] = f1 / fz + np.arange(m) * ((f2 - f1) / (m * fz))   # ZoomFFT(m, [f1, f2], fs=fz)
        return freq

    # ------------------------------------------------------------------ host-side peak picking on the [vel, angle] map
    @staticmethod
    def _floored_db(resp_mag: np.ndarray, min_threshold_dB: float) -> np.ndarray:
        """dB map with everything more than ``min_threshold_dB`` below the global maximum lifted to that floor."""
        db = 20 * np.log10(np.abs(resp_mag) + 1e-12)
        floor = np.max(db) - min_threshold_dB
        db[db <= floor] = floor
        return db

    def detect_peaks_rows(self, doppler_azimuth_resp_mag: np.ndarray, vel_bins: np.ndarray,
                          min_threshold_dB: float = 30.0) -> np.ndarray:
        """One peak per velocity row: the strongest local maximum with at least 4 dB prominence.  ``(N, 2)`` rows of
        (angle in radians, velocity) (:336-389)."""
        db = self._floored_db(doppler_azimuth_resp_mag, min_threshold_dB)
        angles, vels = [], []
        for r, row in enumerate(db):
            cand, _ = find_peaks(row, prominence=4.0)
            if cand.size > 0:
                angles.append(self.valid_angle_bins[cand[np.argmax(row[cand])]])
                vels.append(vel_bins[r])
        return np.stack([np.array(angles), np.array(vels)], axis=1)

    def detect_peak_zero_az(self, doppler_azimuth_resp_mag: np.ndarray, vel_bins: np.ndarray,
                            min_threshold_dB: float = 30.0) -> np.ndarray:
        """Strongest local maximum down the column closest to zero azimuth: ``[0.0, velocity]``, or an empty
        ``(0, 2)`` array when that column has no peak (:392-417)."""
        db = self._floored_db(doppler_azimuth_resp_mag, min_threshold_dB)
        col = db[:, np.argmin(np.abs(self.valid_angle_bins))]
        cand, _ = find_peaks(col)
        if cand.size > 0:
            return np.array([0.0, vel_bins[cand[np.argmax(col[cand])]]])
        return np.empty(shape=(0, 2))

    def process(self, adc_cube: np.ndarray, rx_antennas: Union[np.ndarray, list] = [],
                range_window: Union[np.ndarray, list] = [], shift_angle: bool = True, use_precise_fft: bool = False,
                precise_vel_range=np.array([-0.25, 0.25]), **kwargs) -> np.ndarray:
        """float64 ``[vel bins, valid angle bins]``, averaged over the range bins inside ``range_window``."""
        rx = np.array([]) if rx_antennas is None else np.asarray(rx_antennas).astype(int).ravel()
        rw = np.array([]) if range_window is None else np.asarray(range_window, dtype=float).ravel()
        cube = np.asarray(adc_cube)
        if rx.size > 0:
            cube = cube[rx, :, :]          # subset FIRST: the antenna window spans the selected antennas (:462-466)
        cm = self.config_manager
        if rw.size == 0:
            rw = np.array([0, cm.range_max_m])
        keep = np.where((self.range_bins >= rw[0]) & (self.range_bins <= rw[1]))[0]
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(cube)
        A = int(self.num_angle_bins)
        if S != len(self.range_bins) or A < V:
            raise ValueError(f"cube {cube.shape} inconsistent with cfg ({len(self.range_bins)} range bins) / {A} angle bins")
        if keep.size == 0:                 # np.mean over an empty axis: NaNs, like the reference
            return np.full((C, int(self.valid_angle_mask.sum())), np.nan)
        flags = _lib.ANGLE_MAGNITUDE
        if not (cm.array_geometry == "standard" and cm.virtual_antennas_enabled):
            flags |= _lib.ANGLE_NO_WINDOW
        if not shift_angle:
            flags |= _lib.ANGLE_NO_SHIFT
        if use_precise_fft:
            vr = np.array([-0.25, 0.25]) if precise_vel_range is None else precise_vel_range
            freq = np.ascontiguousarray(self._zoom_plan(vr), dtype=np.float64)
            M = int(freq.size)
            n_used = int(self.vel_bins.size)        # zoom_fft keeps the first num_samples chirps (:159-160)
            if M == 0:
                return np.zeros((0, int(self.valid_angle_mask.sum())))
            if n_used > C:
                raise ValueError(f"CZT defined for length {n_used}, not {C}")     # scipy's error for the same input
            d_out = bufs.get("da_out", M * A * 4)
            _lib.check(ctx.lib.mmw_doppler_azimuth_zoom(
                ctx.handle, d_cube.ptr, d_out.ptr, 1, V, S, C, A, int(keep[0]), int(keep[-1]) + 1, n_used,
                freq.ctypes.data_as(_lib.C.POINTER(_lib.C.c_double)), M,
                flags & (_lib.ANGLE_NO_WINDOW | _lib.ANGLE_NO_SHIFT)))
            resp = d_out.download((M, A), np.float32).astype(np.float64)
            return resp[:, self.valid_angle_mask]
        d_out = bufs.get("da_out", C * A * 4)
        _lib.check(ctx.lib.mmw_doppler_azimuth(ctx.handle, d_cube.ptr, d_out.ptr, 1, V, S, C, A, int(keep[0]),
                                               int(keep[-1]) + 1, flags & (_lib.ANGLE_NO_WINDOW | _lib.ANGLE_NO_SHIFT)))
        resp = d_out.download((C, A), np.float32).astype(np.float64)
        return resp[:, self.valid_angle_mask]
