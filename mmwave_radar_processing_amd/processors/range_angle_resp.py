"""Range-angle response of one chirp (reference: mmwave_radar_processing/processors/range_angle_resp.py:6-122)."""
from __future__ import annotations

import numpy as np

from .. import _lib
from ._processor import _Processor


def angle_tables(num_angle_bins: int):
    """phase_shifts / angle_bins tables (range_angle_resp.py:38-48): spacing 2*pi/(A-1), last forced to -pi."""
    step = 2 * np.pi / (num_angle_bins - 1)
    phase = np.arange(start=np.pi, stop=-np.pi - step, step=-step)
    phase[-1] = -1 * np.pi
    return phase, np.arcsin(phase / np.pi)


class RangeAngleProcessor(_Processor):
    def __init__(self, config_manager, num_angle_bins: int = 64, **kwargs) -> None:
        self.num_angle_bins = num_angle_bins
        self.phase_shifts = None
        self.angle_bins = None
        self.num_range_bins = None
        self.range_bins = None
        super().__init__(config_manager)

    def _range_bins_offset(self):
        cm = self.config_manager
        return np.arange(start=0, step=cm.range_res_m, stop=cm.range_max_m - cm.range_res_m / 2) + 1e-3

    def _mesh(self, angle_bins):
        self.angle_bins = angle_bins
        self.thetas, self.rhos = np.meshgrid(angle_bins, self.range_bins)
        self.x_s = np.multiply(self.rhos, np.cos(self.thetas))
        self.y_s = np.multiply(self.rhos, np.sin(self.thetas))

    def configure(self):
        cm = self.config_manager
        self.num_range_bins = cm.get_num_adc_samples(profile_idx=0)
        self.range_bins = self._range_bins_offset()          # +1 mm, unlike the RD processor (:31-34)
        self.num_rx_antennas = cm.num_rx_antennas
        self.phase_shifts, bins = angle_tables(self.num_angle_bins)
        self._mesh(bins)

    def process(self, adc_cube: np.ndarray, chirp_idx=0, rx_antennas: np.ndarray = np.array([]),
                perform_windowing: bool = True, **kwargs) -> np.ndarray:
        """|fftshift_A fft2(zero-pad_A((hann(S) hann(V) x)[rx, :, chirp].T))| -> float64 (S, A).

        The antenna window spans ALL V antennas before the subset is taken (reference :96-101)."""
        rx = np.asarray(rx_antennas).astype(int).ravel()
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        if S != self.num_range_bins:
            raise ValueError(f"cube has {S} samples, cfg says {self.num_range_bins}")
        A = int(self.num_angle_bins)
        d_out = bufs.get("ra", S * A * 4)
        arr, n = _lib.int_array(rx)
        _lib.check(ctx.lib.mmw_range_angle(ctx.handle, d_cube.ptr, d_out.ptr, 1, V, S, C, A, int(chirp_idx),
                                           arr, n, int(bool(perform_windowing))))
        return d_out.download((S, A), np.float32).astype(np.float64)
