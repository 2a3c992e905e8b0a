"""Range-angle response with Doppler beam sharpening
(reference: mmwave_radar_processing/processors/range_angle_resp_dbs_enhanced.py:7-342).

``compute_3d_windowed_fft`` is the headline chain: Hann-windowed range FFT ->
Doppler FFT + fftshift -> Hann(V), zero-pad V->A, angle FFT + fftshift, run by
``mmw_chain3d`` (fused range-Doppler kernel + streaming angle kernel).
"""
from __future__ import annotations

import numpy as np

from .. import _lib
from .range_angle_resp import RangeAngleProcessor, angle_tables


class RangeAngleProcessorDBSEnhanced(RangeAngleProcessor):
    def __init__(self, config_manager, num_angle_bins_range_angle_response: int = 64,
                 num_angle_bins_dbs_enhanced_response: int = 64, min_x_y_vel_dbs: float = 0.25, **kwargs) -> None:
        self.angle_bins_no_dbs_enhancement = None
        self.num_angle_bins_dbs_enhanced_response = num_angle_bins_dbs_enhanced_response
        self.angle_bins_dbs_enhanced = None
        self.phase_shifts_dbs_enhanced = None
        self.min_vel_dbs = min_x_y_vel_dbs
        self.vel_bins = None
        super().__init__(config_manager=config_manager, num_angle_bins=num_angle_bins_range_angle_response, **kwargs)

    def configure(self) -> None:
        cm = self.config_manager
        self.num_range_bins = cm.get_num_adc_samples(profile_idx=0)
        self.range_bins = self._range_bins_offset()
        self.vel_bins = np.arange(start=-1 * cm.vel_max_m_s, stop=cm.vel_max_m_s - cm.vel_res_m_s + 1e-3,
                                  step=cm.vel_res_m_s)
        self.num_rx_antennas = cm.num_rx_antennas
        self.phase_shifts, self.angle_bins_no_dbs_enhancement = angle_tables(self.num_angle_bins)
        self.angle_bins_dbs_enhanced = np.linspace(start=self.angle_bins_no_dbs_enhancement[0],
                                                   stop=self.angle_bins_no_dbs_enhancement[-1],
                                                   num=self.num_angle_bins_dbs_enhanced_response)
        self.compute_mesh_grid()

    def compute_mesh_grid_dbs_enhanced(self) -> None:
        self._mesh(self.angle_bins_dbs_enhanced)

    def compute_mesh_grid(self) -> None:
        self._mesh(self.angle_bins_no_dbs_enhancement)

    def process_no_dbs(self, adc_cube, chirp_idx=0, rx_antennas=np.array([]), **kwargs) -> np.ndarray:
        self.compute_mesh_grid()
        return RangeAngleProcessor.process(self, adc_cube=adc_cube, chirp_idx=chirp_idx, rx_antennas=rx_antennas,
                                           **kwargs)

    # ------------------------------------------------------------------ the 3-D chain
    def _chain3d_device(self, adc_cube, magnitude: bool):
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        A = int(self.num_angle_bins)
        if A < V:
            raise ValueError(f"num_angle_bins ({A}) must be >= number of antennas ({V})")
        d_out = bufs.get("cube3d", A * S * C * (4 if magnitude else 8))
        _lib.check(ctx.lib.mmw_chain3d(ctx.handle, d_cube.ptr, None, d_out.ptr, 1, V, S, C, A, int(magnitude)))
        return d_out, (A, S, C)

    def compute_3d_windowed_fft(self, adc_cube: np.ndarray) -> np.ndarray:
        """complex128 ``(num_angle_bins, samples, chirps)`` (reference :137-198)."""
        d_out, shape = self._chain3d_device(adc_cube, magnitude=False)
        _, bufs = self._device()
        return d_out.download_widened(shape, np.complex64, staging=bufs.get("widen", int(np.prod(shape)) * 16))

    def get_dop_vel(self, angle: float, ego_vel: np.ndarray) -> float:
        r = np.array([np.cos(angle), np.sin(angle), 0])
        return -1 * np.dot((r / np.linalg.norm(r)), ego_vel)

    def _dbs_indices(self, velocity_ned):
        """Nearest (angle bin, Doppler bin) per sharpened output angle -- host argmin over the bin tables (:240-257)."""
        ang = self.angle_bins_dbs_enhanced
        dop = np.array([self.get_dop_vel(a, velocity_ned) for a in ang])
        vel_bin = np.argmin(np.abs(self.vel_bins[None, :] - dop[:, None]), axis=1)
        ang_bin = np.argmin(np.abs(self.angle_bins_no_dbs_enhancement[None, :] - ang[:, None]), axis=1)
        return ang_bin, vel_bin

    def perform_dbs_sharpen(self, velocity_ned: np.ndarray, angle_rng_dop_resp_mag: np.ndarray) -> np.ndarray:
        """Pick, per output angle, the [nearest angle bin, :, nearest Doppler bin] column of a HOST magnitude cube
        (reference :216-263); ``process_dbs_enhanced`` does the same pick on the device-resident cube."""
        ang_bin, vel_bin = self._dbs_indices(velocity_ned)
        picked = np.asarray(angle_rng_dop_resp_mag)[ang_bin, :, vel_bin]        # (A', S)
        return np.ascontiguousarray(picked.T)

    def process_dbs_enhanced(self, adc_cube, velocity_ned, rx_antennas=np.array([]), **kwargs) -> np.ndarray:
        self.compute_mesh_grid_dbs_enhanced()
        rx = np.asarray(rx_antennas).astype(int).ravel()
        cube = np.asarray(adc_cube)
        if rx.size > 0:
            cube = cube[rx, :, :]
        d_mag, (A, S, C) = self._chain3d_device(cube, magnitude=True)       # |.| on the GPU (:293)
        ang_bin, vel_bin = self._dbs_indices(velocity_ned)
        ctx, bufs = self._device()
        n_out = len(ang_bin)
        d_out = bufs.get("dbs_out", S * n_out * 4)
        a_arr, _ = _lib.int_array(ang_bin)
        v_arr, _ = _lib.int_array(vel_bin)
        _lib.check(ctx.lib.mmw_dbs_gather(ctx.handle, d_mag.ptr, a_arr, v_arr, d_out.ptr, 1, A, S, C, n_out))
        return d_out.download((S, n_out), np.float32).astype(np.float64)

    def process(self, adc_cube, velocity_ned, rx_antennas=np.array([]), chirp_idx: int = 0, **kwargs) -> np.ndarray:
        if np.linalg.norm(np.asarray(velocity_ned)[0:2]) < self.min_vel_dbs:
            return self.process_no_dbs(adc_cube=adc_cube, chirp_idx=chirp_idx, rx_antennas=rx_antennas, **kwargs)
        return self.process_dbs_enhanced(adc_cube=adc_cube, velocity_ned=velocity_ned, rx_antennas=rx_antennas,
                                         **kwargs)
