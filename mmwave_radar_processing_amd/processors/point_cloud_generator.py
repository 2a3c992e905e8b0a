"""Point cloud from raw ADC cubes
(reference: mmwave_radar_processing/processors/point_cloud_generator.py:9-256).

detector (RD + CFAR on the GPU) -> per-detection zero-padded angle FFT + first-max argmax on the GPU
(``mmw_angle_argmax_exact``, one wavefront per detection, float64 where float32 cannot decide) -> host table lookups
and the spherical->Cartesian map.
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
        self.n_refined = 0      # detections of the last frame whose argmax was re-evaluated in float64
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
        det = self.detector
        # detections and both argmax index lists from ONE device call where the detector has the fused kernel
        # (RangeDopplerDetector2D with a CA-CFAR whose plane fits the LDS, lists of <= 8 antennas): mmw_detect_points
        fused = getattr(det, "process_points", None)
        if fused is not None and max(self.az_antenna_idxs.size, self.el_antenna_idxs.size) <= 8:
            dets = fused(adc_cube, [int(i) for i in self.az_antenna_idxs], [int(i) for i in self.el_antenna_idxs],
                         self.shift_az_resp, self.shift_el_resp, self.num_angle_bins)
            if dets is not None:
                self.n_refined = int(det.screen_stats[3] + det.screen_stats[4])
                if dets.shape[0] == 0:
                    return np.empty((0, 4))
                det_ranges, det_velocities, _, _ = det._map_detections_to_bins(dets)
                az_idx, el_idx = det.points
                n = dets.shape[0]
                az = self.angle_bins[az_idx] if az_idx is not None else np.zeros(n)
                el = self.angle_bins[el_idx] if el_idx is not None else np.zeros(n)
                return self._convert_to_cartesian(det_ranges, az, el, det_velocities)
        dets = det.process(adc_cube, **kwargs)
        if dets.shape[0] == 0:
            return np.empty((0, 4))
        det_ranges, det_velocities, r_idx, v_idx = det._map_detections_to_bins(dets)
        # (None = "the cube the detector has just left on the device": passing detector.rng_dop_resp_raw, as the reference
        #  does, would first download and convert it)
        az, el = self._compute_angle_estimation(None if getattr(det, "_dev", None) is not None else det.rng_dop_resp_raw,
                                                r_idx, v_idx)
        return self._convert_to_cartesian(det_ranges, az, el, det_velocities)

    # ------------------------------------------------------------------ angles
    def _compute_angle_estimation(self, rng_dop_resp_raw: np.ndarray, det_range_idxs: np.ndarray,
                                  det_velocity_idxs: np.ndarray):
        """Azimuth / elevation angle of each detection; an empty antenna list gives zeros (reference :160-178).

        The reference does the per-detection angle FFT in complex128 on a complex128 range-Doppler cube.  Two device
        paths give the same argmax indices:
          * the cube is the one the detector just computed (still resident in HBM, float32): ``mmw_angle_argmax_exact``
            -- float32 pass with an error bound, detections whose two best bins are closer than the bound are
            re-evaluated in float64 from the raw ADC cube;
          * any other array (a caller's own complex128 cube): the cells at the detections are gathered on the host and
            the float64 angle DFT + argmax runs on the device (``mmw_angle_argmax_cells64``)."""
        n = len(det_range_idxs)
        az_angles = np.zeros(n)
        el_angles = np.zeros(n)
        if n == 0 or (self.az_antenna_idxs.size == 0 and self.el_antenna_idxs.size == 0):
            return az_angles, el_angles
        ctx, bufs = self._device()
        L, h = ctx.lib, ctx.handle
        r_idx = np.asarray(det_range_idxs).astype(np.int64)
        v_idx = np.asarray(det_velocity_idxs).astype(np.int64)
        A = int(self.num_angle_bins)
        d_idx = bufs.get("angle_idx", n * 4)
        dev = getattr(self.detector, "_dev", None)
        resident = dev is not None and (rng_dop_resp_raw is None or rng_dop_resp_raw is self.detector.__dict__.get("rng_dop_resp_raw"))
        raw = None if resident else np.asarray(rng_dop_resp_raw)
        if resident:
            d_rd, (V, S, C), d_cube = dev[0], dev[2], dev[3]
            dets = np.ascontiguousarray(np.stack([r_idx, v_idx], axis=1), dtype=np.int32)
            d_dets, d_cnt = bufs.get("pc_dets", dets.nbytes), bufs.get("pc_count", 4)
            d_dets.upload(dets)
            d_cnt.upload(np.array([n], dtype=np.int32))
            d_l1 = bufs.get("plane_l1", V * 4)
            _lib.check(L.mmw_plane_l1(h, d_cube.ptr, d_l1.ptr, 1, V, S, C))
        self.n_refined = 0
        out = []
        for ant, shift in ((self.az_antenna_idxs, self.shift_az_resp), (self.el_antenna_idxs, self.shift_el_resp)):
            if ant.size == 0:
                out.append(np.zeros(n))
                continue
            if resident:
                arr, n_ant = _lib.int_array(ant)
                n_ref = _lib.C.c_int(0)
                _lib.check(L.mmw_angle_argmax_exact(h, d_cube.ptr, d_l1.ptr, d_rd.ptr, d_dets.ptr, d_cnt.ptr, d_idx.ptr, 1,
                                                    V, S, C, n, arr, n_ant, A, int(bool(shift)), _lib.C.byref(n_ref)))
                self.n_refined += n_ref.value
            else:
                cells = np.ascontiguousarray(raw[ant][:, r_idx, v_idx].T, dtype=np.complex128)     # (N, n_ant), :168-175
                d_cells = bufs.get("pc_cells", cells.nbytes)
                d_cells.upload(cells)
                _lib.check(L.mmw_angle_argmax_cells64(h, d_cells.ptr, d_idx.ptr, n, cells.shape[1], A, int(bool(shift))))
            out.append(self.angle_bins[d_idx.download((n,), np.int32).astype(int)])
        return out[0], out[1]

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
