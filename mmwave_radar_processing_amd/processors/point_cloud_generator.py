"""Point cloud from raw ADC cubes
(reference: mmwave_radar_processing/processors/point_cloud_generator.py:9-256).

detector (RD + CFAR on the GPU) -> per-detection zero-padded angle FFT + first-max argmax on the GPU
(``mmw_angle_argmax``, one wavefront per detection) -> host table lookups and the spherical->Cartesian map.
"""
from __future__ import annotations

from typing import Dict, List, Union

import numpy as np

from .. import _lib
from ._processor import _Processor
from .range_angle_resp import angle_tables
from .range_doppler_detection.registry import get_range_doppler_detector_registry


def _as_index_array(idxs, what):
    if idxs is None:
        return np.array([], dtype=int)
    if isinstance(idxs, (list, tuple)):
        return np.array(idxs, dtype=int)
    if isinstance(idxs, np.ndarray):
        return idxs.astype(int)
    raise ValueError(f"{what} must be a list or numpy array")


class PointCloudGenerator(_Processor):
    def __init__(self, config_manager, az_antenna_idxs: Union[List[int], np.ndarray],
                 el_antenna_idxs: Union[List[int], np.ndarray], detector_type: str = "range_doppler_detector_2d",
                 detector_params: Dict = {}, shift_az_resp: bool = True, shift_el_resp: bool = False,
                 num_angle_bins: int = 64, **kwargs):
        self.shift_az_resp = shift_az_resp
        self.shift_el_resp = shift_el_resp
        self.az_antenna_idxs = _as_index_array(az_antenna_idxs, "az_antenna_idxs")
        self.el_antenna_idxs = _as_index_array(el_antenna_idxs, "el_antenna_idxs")
        self.num_angle_bins = num_angle_bins
        self.phase_shifts = None
        self.angle_bins = None
        registry = get_range_doppler_detector_registry()
        if detector_type not in registry:
            raise ValueError(f"Unknown detector type: {detector_type}. Available: {list(registry.keys())}")
        self.detector = registry[detector_type](config_manager, **detector_params)
        super().__init__(config_manager)
        self.logger.info(f"PointCloudGenerator initialized with detector: {detector_type}")

    def configure(self):
        self.detector.configure()
        self.phase_shifts, self.angle_bins = angle_tables(self.num_angle_bins)

    def process(self, adc_cube: np.ndarray, **kwargs) -> np.ndarray:
        dets = self.detector.process(adc_cube, **kwargs)
        if dets.shape[0] == 0:
            return np.empty((0, 4))
        det_ranges, det_velocities, r_idx, v_idx = self.detector._map_detections_to_bins(dets)
        az, el = self._compute_angle_estimation(self.detector.rng_dop_resp_raw, r_idx, v_idx)
        return self._convert_to_cartesian(det_ranges, az, el, det_velocities)

    # ------------------------------------------------------------------ angles
    def _argmax_bins(self, ctx, bufs, d_rd_ptr, shape, d_dets, d_cnt, cap, n, ant, shift):
        V, S, C = shape
        d_idx = bufs.get("angle_idx", max(cap, 1) * 4)
        arr, n_ant = _lib.int_array(ant)
        _lib.check(ctx.lib.mmw_angle_argmax(ctx.handle, d_rd_ptr, d_dets.ptr, d_cnt.ptr, d_idx.ptr, 1, V, S, C, cap,
                                            arr, n_ant, int(self.num_angle_bins), int(bool(shift))))
        return d_idx.download((n,), np.int32).astype(int)

    def _compute_angle_estimation(self, rng_dop_resp_raw: np.ndarray, det_range_idxs: np.ndarray,
                                  det_velocity_idxs: np.ndarray):
        """Azimuth / elevation angle of each detection; an empty antenna list gives zeros (reference :160-178)."""
        n = len(det_range_idxs)
        az_angles = np.zeros(n)
        el_angles = np.zeros(n)
        if n == 0 or (self.az_antenna_idxs.size == 0 and self.el_antenna_idxs.size == 0):
            return az_angles, el_angles
        ctx, bufs = self._device()
        raw = np.asarray(rng_dop_resp_raw)
        dev = getattr(self.detector, "_dev", None)
        if dev is not None and raw is self.detector.rng_dop_resp_raw:
            d_rd, shape = dev[0], dev[2]          # RD cube of this frame is still resident in HBM
        else:
            rd = np.ascontiguousarray(raw, dtype=np.complex64)
            d_rd = bufs.get("rd_host", rd.nbytes)
            d_rd.upload(rd)
            shape = rd.shape
        dets = np.ascontiguousarray(np.stack([np.asarray(det_range_idxs), np.asarray(det_velocity_idxs)], axis=1),
                                    dtype=np.int32)
        d_dets, d_cnt = bufs.get("pc_dets", dets.nbytes), bufs.get("pc_count", 4)
        d_dets.upload(dets)
        d_cnt.upload(np.array([n], dtype=np.int32))
        if self.az_antenna_idxs.size > 0:
            idx = self._argmax_bins(ctx, bufs, d_rd.ptr, shape, d_dets, d_cnt, n, n, self.az_antenna_idxs,
                                    self.shift_az_resp)
            az_angles = self.angle_bins[idx]
        if self.el_antenna_idxs.size > 0:
            idx = self._argmax_bins(ctx, bufs, d_rd.ptr, shape, d_dets, d_cnt, n, n, self.el_antenna_idxs,
                                    self.shift_el_resp)
            el_angles = self.angle_bins[idx]
        return az_angles, el_angles

    def _convert_to_cartesian(self, ranges, az_angles, el_angles, velocities) -> np.ndarray:
        """FLU frame: x forward, y left, z up (reference :216-248).  O(N) host arithmetic on the detections."""
        cos_el = np.cos(el_angles)
        x = ranges * cos_el * np.cos(az_angles)
        y = ranges * cos_el * np.sin(az_angles)
        z = ranges * np.sin(el_angles)
        return np.column_stack((x, y, z, velocities))

    def reset(self):
        self.detector.reset()
        return super().reset()
