"""Strip-map SAR image of one frame (reference: mmwave_radar_processing/processors/strip_map_SAR_processor.py:9-196).

The image is the un-windowed 2-D FFT of one antenna's (sample x chirp) plane, fftshifted along the chirp axis -- complex128
on the device (``mmw_fft2_c128``) -- cut down to the ground patch the platform's motion defines.  The geometry (cross-range
angle of every chirp phase step from the platform velocity, slant -> ground range) is host metadata, restated expression by
expression so that the bin tables are bit-identical."""
from __future__ import annotations

import numpy as np
import scipy.constants as constants

from .. import _lib
from ._processor import _Processor
from .virtual_array_reformater import VirtualArrayReformatter


class StripMapSARProcessor(_Processor):
    def __init__(self, config_manager, az_angle_range_rad=np.deg2rad(np.array([-30, 30])), **kwargs):
        if isinstance(az_angle_range_rad, list):
            az_angle_range_rad = np.array(az_angle_range_rad)
        if config_manager.virtual_antennas_enabled:
            self.virtual_array_reformatter = VirtualArrayReformatter(config_manager)
        self.chirps_per_frame = None
        self.chirp_period_us = None
        self.chirp_tx_masks = None
        self.chirp_rx_positions_m = None
        self.lambda_m = None
        self.phase_shifts = None
        self.angle_bins_rad = None
        self.az_angle_range_rad = az_angle_range_rad
        self.num_range_bins = None
        self.radar_range_bins = None
        self.valid_ranges_slice = None
        self.valid_angles_slice = None
        self.ground_range_bins = None
        self.ground_az_bins_rad = None
        self.thetas = None
        self.rhos = None
        self.x_s = None
        self.y_s = None
        super().__init__(config_manager)

    def configure(self):
        self._compute_key_radar_parameters()

    def _compute_key_radar_parameters(self):
        cm = self.config_manager
        self.num_range_bins = cm.get_num_adc_samples(profile_idx=0)
        self.range_bins = np.linspace(start=0, stop=cm.range_max_m, num=self.num_range_bins)
        self.lambda_m = constants.c / (float(cm.profile_cfgs[0]["startFreq_GHz"]) * 1e9)
        self.chirps_per_frame = cm.frameCfg_loops * (cm.frameCfg_end_index - cm.frameCfg_start_index + 1)
        self.phase_shifts = np.linspace(start=np.pi, stop=-np.pi, num=cm.frameCfg_loops)       # one per chirp loop
        self.chirp_period_us = cm.profile_cfgs[0]["idleTime_us"] + cm.profile_cfgs[0]["rampEndTime_us"]

    def configure_array_geometry(self, vel_m_per_s: float, sensor_height_m: float, max_SAR_distance: float):
        """Ground patch for this platform speed (:97-150): synthetic element spacing, cross-range angle per Doppler bin, the
        angle and slant-range windows, and the polar / Cartesian ground grids."""
        d_rx = 2 * self.chirp_period_us * 1e-6 * vel_m_per_s
        self.angle_bins_rad = np.arcsin(self.phase_shifts * self.lambda_m) / (2 * np.pi * d_rx)
        edges = [int(np.argmin(np.abs(self.angle_bins_rad - lim)))
                 for lim in (np.min(self.az_angle_range_rad), np.max(self.az_angle_range_rad))]
        self.valid_angles_slice = slice(np.min(edges), np.max(edges))
        self.ground_az_bins_rad = self.angle_bins_rad[self.valid_angles_slice]
        first = np.nonzero(self.range_bins > sensor_height_m)[0][0]
        last = np.nonzero(self.range_bins < max_SAR_distance)[0][-1]
        self.valid_ranges_slice = slice(first, last)
        self.ground_range_bins = np.sqrt(np.power(self.range_bins[self.valid_ranges_slice], 2) - np.power(d_rx, 2))
        self.thetas, self.rhos = np.meshgrid(self.ground_az_bins_rad, self.ground_range_bins, indexing='xy')
        self.x_s = np.multiply(self.rhos, np.cos(self.thetas))
        self.y_s = np.multiply(self.rhos, np.sin(self.thetas))

    def process(self, adc_cube, vel_m_per_s: float, sensor_height_m: float = 0.24, rx_index=0,
                max_SAR_distance: float = 1.5, **kwargs):
        if self.config_manager.virtual_antennas_enabled:
            adc_cube = self.virtual_array_reformatter.process(adc_cube=adc_cube)
        self.configure_array_geometry(vel_m_per_s=vel_m_per_s, sensor_height_m=sensor_height_m,
                                      max_SAR_distance=max_SAR_distance)
        ctx, bufs, d_cube, (V, S, C) = self._upload_cube(adc_cube)
        rx = int(rx_index) + V if int(rx_index) < 0 else int(rx_index)
        d_img = bufs.get("fft2_c128", S * C * 16)
        _lib.check(ctx.lib.mmw_fft2_c128(ctx.handle, d_cube.ptr, d_img.ptr, 1, V, S, C, rx))
        image = d_img.download((S, C), np.complex128)
        return image[self.valid_ranges_slice, self.valid_angles_slice]
