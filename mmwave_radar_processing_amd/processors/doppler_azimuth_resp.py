"""Doppler-azimuth response, coarse FFT path
(reference: mmwave_radar_processing/processors/doppler_azimuth_resp.py:8-128,296-334,419-491).

antenna subset -> Hann(S) x Hann(C) (x Hann(V) on the "standard" geometry with virtual antennas) -> range FFT ->
keep the range bins inside ``range_window`` -> 2-D FFT over (chirp, zero-padded antenna) -> fftshift -> |.| ->
valid angle columns -> mean over the kept range bins.  On the GPU this is the 3-D chain of
``mmw_chain3d`` (magnitude output) followed by ``mmw_mean_over_range``.
The reference's ZoomFFT ("precise") mode and scipy peak pickers are a later-round item.
"""
from __future__ import annotations

from typing import Union

import numpy as np

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

    def process(self, adc_cube: np.ndarray, rx_antennas: Union[np.ndarray, list] = [],
                range_window: Union[np.ndarray, list] = [], shift_angle: bool = True, use_precise_fft: bool = False,
                precise_vel_range=np.array([-0.25, 0.25]), **kwargs) -> np.ndarray:
        """float64 ``[vel bins, valid angle bins]``, averaged over the range bins inside ``range_window``."""
        if use_precise_fft:
            raise NotImplementedError("the ZoomFFT (precise) Doppler-azimuth mode is not accelerated yet")
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
        d_mag = bufs.get("da_mag", A * S * C * 4)
        d_out = bufs.get("da_out", C * A * 4)
        _lib.check(ctx.lib.mmw_chain3d(ctx.handle, d_cube.ptr, None, d_mag.ptr, 1, V, S, C, A, flags))
        _lib.check(ctx.lib.mmw_mean_over_range(ctx.handle, d_mag.ptr, d_out.ptr, 1, A, S, C, int(keep[0]),
                                               int(keep[-1]) + 1))
        resp = d_out.download((C, A), np.float32).astype(np.float64)
        return resp[:, self.valid_angle_mask]
