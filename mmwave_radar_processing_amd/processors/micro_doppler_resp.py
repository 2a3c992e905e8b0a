"""Micro-Doppler spectrogram (reference: mmwave_radar_processing/processors/micro_doppler_resp.py:6-114).

Per frame: the un-windowed 2-D FFT magnitude of one antenna (``mmw_fft2_mag64``, float64 on the device), its maximum over
the range bins of the target window, pushed into a rolling (velocity x frames) history.  The history is state across
frames, so the class lives on the single-frame API only."""
from __future__ import annotations

import numpy as np

from .. import _lib
from ._processor import _Processor


class MicroDopplerProcessor(_Processor):
    def __init__(self, config_manager, target_ranges=[0, 1.0], num_frames_history: int = 20, **kwargs) -> None:
        if isinstance(target_ranges, list):
            target_ranges = np.array(target_ranges)
        self.vel_bins = None
        self.range_bins = None
        self.target_ranges = target_ranges
        self.range_bin_idxs_to_keep = None
        self.num_frames_history = num_frames_history
        self.time_bins = None
        self.micro_doppler_resp = None
        super().__init__(config_manager)

    def reset(self):
        self.micro_doppler_resp = np.zeros(shape=(self.vel_bins.shape[0], self.num_frames_history))
        super().reset()

    def configure(self):
        cm = self.config_manager
        self.vel_bins = np.arange(start=-1 * cm.vel_max_m_s, stop=cm.vel_max_m_s - cm.vel_res_m_s + 1e-3, step=cm.vel_res_m_s)
        self.range_bins = np.arange(start=0, step=cm.range_res_m, stop=cm.range_max_m - cm.range_res_m / 2 + 1e-3)
        self.range_bin_idxs_to_keep = np.logical_and(self.range_bins >= self.target_ranges[0],
                                                     self.range_bins <= self.target_ranges[1]).astype(np.bool_)
        self.micro_doppler_resp = np.zeros(shape=(self.vel_bins.shape[0], self.num_frames_history))
        frame_period = cm.frameCfg_periodicity_ms * 1e-3
        self.time_bins = np.linspace(start=0, stop=self.num_frames_history * frame_period, num=self.num_frames_history)

    def process(self, adc_cube: np.ndarray, rx_idx=0, **kwargs) -> np.ndarray:
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        rx = int(rx_idx) + V if int(rx_idx) < 0 else int(rx_idx)
        d_mag = bufs.get("fft2_mag64", S * C * 8)
        _lib.check(ctx.lib.mmw_fft2_mag64(ctx.handle, d_cube.ptr, d_mag.ptr, 1, V, S, C, rx))
        response = d_mag.download((S, C), np.float64)
        slice_to_keep = np.max(response[self.range_bin_idxs_to_keep, :], axis=0)
        self.micro_doppler_resp[:, 1:] = self.micro_doppler_resp[:, 0:-1]
        self.micro_doppler_resp[:, 0] = slice_to_keep
        return self.micro_doppler_resp
