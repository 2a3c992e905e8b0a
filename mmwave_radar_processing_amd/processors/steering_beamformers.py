"""Steering-matrix beamformers on the MI355X matrix cores.

``SyntheticArrayBeamformerCore`` is the dense contraction of the reference's
``SyntheticArrayBeamformerProcessor`` (mmwave_radar_processing/processors/
simple_synthetic_array_beamformer_processor_multiFrame.py:474-585): steering directions, Hamming over array
elements, phase-steer-and-sum, Hann over samples, range FFT.  The velocity-history / geometry bookkeeping
around it is sequential host state that SURVEY.md keeps out of scope; callers hand in the stacked history
``[frames, S, chirps]`` and geometry ``[frames, 3, chirps]`` exactly as the reference stores them.

``CaponBeamformer`` is the MVDR spectrum BASELINE.json config 4 asks for.  The reference has NO Capon code
(SURVEY.md F2): the definition is this build's own and its parity is "unpinned".
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib


class SyntheticArrayBeamformerCore:
    def __init__(self, az_angle_bins_rad, el_angle_bins_rad, lambda_m: float, ctx: _lib.Context = None):
        self.az_angle_bins_rad = np.asarray(az_angle_bins_rad, dtype=float)
        self.el_angle_bins_rad = np.asarray(el_angle_bins_rad, dtype=float)
        self.lambda_m = float(lambda_m)
        self._compute_beam_stearing_vectors()
        self._ctx = ctx
        self._bufs = None
        self.beamformed_resp = None

    def _compute_beam_stearing_vectors(self):
        """d[3, n_az, n_el] unit pointing vectors (reference :474-488)."""
        thetas, phis = np.meshgrid(self.az_angle_bins_rad, self.el_angle_bins_rad, indexing="ij")
        self.d = np.array([np.cos(thetas) * np.cos(phis), np.sin(thetas) * np.cos(phis), np.sin(phis)])

    def _device(self):
        if self._ctx is None:
            self._ctx = _lib.default_context()
        if self._bufs is None:
            self._bufs = _lib.BufferSet(self._ctx)
        return self._ctx, self._bufs

    def compute_synthetic_response(self, history_adc_cube: np.ndarray, array_geometry: np.ndarray) -> np.ndarray:
        """history ``[frames, S, chirps]`` complex, geometry ``[frames, 3, chirps]`` -> complex128 ``[S, n_az, n_el]``."""
        hist = np.asarray(history_adc_cube)
        geom = np.asarray(array_geometry, dtype=np.float64)
        if hist.ndim == 2:      # already [S, E]
            X, P = hist, geom.reshape(3, -1)
        else:
            X = hist.transpose((1, 0, 2)).reshape(hist.shape[1], -1)        # [S, E]  (:551-555)
            P = geom.transpose((1, 0, 2)).reshape(3, -1)                    # [3, E]  (:558-559)
        return self.contract(X, P)

    def contract(self, X_se: np.ndarray, P_3e: np.ndarray) -> np.ndarray:
        """``X [S, E]`` with geometry ``P [3, E]`` -> ``[S, n_az, n_el]``; a batch ``X [F, S, E]``, ``P [F, 3, E]`` (every
        frame its own synthetic-array geometry) -> ``[F, S, n_az, n_el]`` in one launch group."""
        ctx, bufs = self._device()
        X = np.ascontiguousarray(X_se, dtype=np.complex64)
        P = np.ascontiguousarray(P_3e, dtype=np.float64)
        single = X.ndim == 2
        if single:
            X, P = X[None], P[None]
        if X.ndim != 3 or P.ndim != 3:
            raise ValueError("expected X [S, E] / P [3, E] or X [F, S, E] / P [F, 3, E]")
        F, S, E = X.shape
        if P.shape != (F, 3, E):
            raise ValueError(f"array geometry {P.shape} does not match {F} frames x {E} elements")
        dirs = np.ascontiguousarray(self.d.reshape(3, -1), dtype=np.float64)
        T = dirs.shape[1]
        d_X, d_P, d_D = bufs.get("bf_x", X.nbytes), bufs.get("bf_p", P.nbytes), bufs.get("bf_d", dirs.nbytes)
        d_Y = bufs.get("bf_y", F * S * T * 8)
        d_X.upload(X)
        d_P.upload(P)
        d_D.upload(dirs)
        _lib.check(ctx.lib.mmw_bartlett(ctx.handle, d_X.ptr, d_P.ptr, d_D.ptr, d_Y.ptr, F, S, E, T, self.lambda_m))
        out = d_Y.download((F, S, T), np.complex64).astype(np.complex128).reshape(F, S, self.d.shape[1], self.d.shape[2])
        self.beamformed_resp = out[0] if single else out
        return self.beamformed_resp


class CaponBeamformer:
    """MVDR angle spectrum per range bin on a V-element half-wavelength ULA (no upstream oracle).

    ``R_r = X_r X_r^H / K + delta tr(R_r)/V I``, ``P(r, theta) = 1 / Re(a^H R_r^-1 a)``,
    ``a_v(theta) = exp(-j pi v sin(theta))``.  float64 on the f64 matrix cores."""

    def __init__(self, thetas_rad, delta: float = 1e-3, ctx: _lib.Context = None):
        self.thetas_rad = np.ascontiguousarray(thetas_rad, dtype=np.float64)
        self.delta = float(delta)
        self._ctx = ctx
        self._bufs = None

    def process(self, X_vrk: np.ndarray) -> np.ndarray:
        """X ``[V, R, K]`` complex snapshots (e.g. range-FFT output per chirp) -> float64 ``[R, T]``;
        a batch ``[F, V, R, K]`` -> ``[F, R, T]`` in one launch."""
        if self._ctx is None:
            self._ctx = _lib.default_context()
        if self._bufs is None:
            self._bufs = _lib.BufferSet(self._ctx)
        ctx, bufs = self._ctx, self._bufs
        X = np.ascontiguousarray(X_vrk, dtype=np.complex64)
        single = X.ndim == 3
        if single:
            X = X[None]
        if X.ndim != 4:
            raise ValueError("expected [V, R, K] or [F, V, R, K] snapshots")
        F, V, R, K = X.shape
        T = len(self.thetas_rad)
        d_X, d_out = bufs.get("cap_x", X.nbytes), bufs.get("cap_p", F * R * T * 4)
        d_X.upload(X)
        th = self.thetas_rad.ctypes.data_as(C.POINTER(C.c_double))
        _lib.check(ctx.lib.mmw_capon(ctx.handle, d_X.ptr, th, d_out.ptr, F, V, R, K, T, self.delta))
        out = d_out.download((F, R, T), np.float32).astype(np.float64)
        return out[0] if single else out
