"""CFAR detector bases: host-side mirror of the reference's detector protocol, compute on MI355X.

Same constructor arguments, ``detect()`` return types and cached attributes
(``thresholds``, ``noise_estimates``, ``detections``) as the reference
(mmwave_radar_processing/detectors/base.py:20-65,185-230); the sliding-window
arithmetic runs in float64 HIP kernels (csrc/mmw_cfar.h) through the C ABI
(``mmw_cfar1d`` / ``mmw_cfar2d``).  Subclasses only describe WHICH statistic
to use (kind, scale factor, order-statistic rank).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

from .. import _lib
from .._lazy import LazyAttrs


def compute_alpha_ca(num_train_cells: int, pfa: float) -> float:
    """alpha = N (Pfa^(-1/N) - 1)  (reference: detectors/base.py:154-169,281-293)."""
    return num_train_cells * (pfa ** (-1.0 / num_train_cells) - 1.0)


class _DeviceCFAR(LazyAttrs):
    """Shared device plumbing: run one CFAR kernel over a float64 host array."""

    _ctx: Optional[_lib.Context] = None
    _bufs: Optional[_lib.BufferSet] = None

    def _context(self) -> _lib.Context:
        if self._ctx is None:
            self._ctx = _lib.default_context()
            self._bufs = _lib.BufferSet(self._ctx)
        return self._ctx

    def _run(self, x: np.ndarray, launch) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        ctx = self._context()
        x = np.ascontiguousarray(x, dtype=np.float64)
        n = x.size
        d_x = self._bufs.get("x", n * 8)
        d_t = self._bufs.get("thr", n * 8)
        d_n = self._bufs.get("noise", n * 8)
        d_m = self._bufs.get("mask", n)
        d_x.upload(x)
        launch(ctx, d_x, d_t, d_n, d_m)
        thr = d_t.download(x.shape, np.float64)
        noise = d_n.download(x.shape, np.float64)
        mask = d_m.download(x.shape, np.uint8).astype(bool)
        return thr, noise, mask


class BaseCFAR1D(_DeviceCFAR):
    """1-D CFAR over a magnitude vector; ``detect`` returns a list of indices."""

    kind = _lib.CFAR_CA

    def __init__(self, num_train: int, num_guard: int, pfa: float, **kwargs):
        self.num_train = num_train
        self.num_guard = num_guard
        self.pfa = pfa
        self.thresholds: Optional[np.ndarray] = None
        self.detections: Optional[np.ndarray] = None
        self.noise_estimates: Optional[np.ndarray] = None

    compute_alpha_ca = staticmethod(compute_alpha_ca)

    # hooks -----------------------------------------------------------
    def _scale(self) -> float:
        raise NotImplementedError

    def _k_rank(self) -> int:
        return 0

    # -----------------------------------------------------------------
    def _compute_thresholds(self, x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        thr, noise, self._mask = self._run_rows(np.asarray(x)[None, :])
        return thr[0], noise[0]

    def _run_rows(self, rows: np.ndarray):
        """rows: (n_rows, L) float64 -> thresholds, noise, mask of the same shape."""
        n_rows, L = rows.shape

        def launch(ctx, d_x, d_t, d_n, d_m):
            _lib.check(ctx.lib.mmw_cfar1d(ctx.handle, d_x.ptr, d_t.ptr, d_n.ptr, d_m.ptr, n_rows, L, self.kind,
                                          int(self.num_train), int(self.num_guard), float(self._scale()),
                                          int(self._k_rank())))
        return self._run(rows, launch)

    def detect(self, x: np.ndarray) -> List[int]:
        x = np.asarray(x)
        if x.ndim != 1:
            raise ValueError("Input x must be a 1D array.")
        # _compute_thresholds is the hook subclasses override (reference base.py:56-65); the decision rule is applied
        # here exactly as the reference does (strict >; +inf / NaN thresholds never fire), so a user subclass written
        # against the reference -- its own NumPy _compute_thresholds -- keeps working unchanged
        self.thresholds, self.noise_estimates = self._compute_thresholds(x)
        self.detections = x > self.thresholds
        return np.where(self.detections)[0].tolist()

    def _get_window_view(self, x: np.ndarray) -> np.ndarray:
        """Sliding windows of 2 * (num_train + num_guard) + 1 cells (reference base.py:129-152): a helper for subclasses
        that compute their own thresholds on the host; ValueError when the input is shorter than the window."""
        window_size = 2 * (self.num_train + self.num_guard) + 1
        if len(x) < window_size:
            raise ValueError(f"Input length {len(x)} is smaller than window size {window_size}.")
        return sliding_window_view(x, window_shape=window_size)

    def plot_detections(self, x, title="CFAR Detection", convert_to_dB=False):  # pragma: no cover
        raise NotImplementedError("plotting is outside the accelerated hot path (SURVEY.md section 2)")


class BaseCFAR2D(_DeviceCFAR):
    """2-D CFAR over a range-Doppler magnitude map; ``detect`` returns (row, col) tuples in row-major order."""

    kind = _lib.CFAR_CA

    def __init__(self, num_train: Tuple[int, int], num_guard: Tuple[int, int], pfa: float, **kwargs):
        self.num_train = num_train          # tuples and YAML lists both accepted (processor_params.yaml:44-45)
        self.num_guard = num_guard
        self.pfa = pfa
        self.thresholds: Optional[np.ndarray] = None
        self.detections: Optional[np.ndarray] = None
        self.noise_estimates: Optional[np.ndarray] = None

    compute_alpha_ca = staticmethod(compute_alpha_ca)

    def num_train_cells(self) -> int:
        (tr, td), (gr, gd) = self.num_train, self.num_guard
        return (2 * (tr + gr) + 1) * (2 * (td + gd) + 1) - (2 * gr + 1) * (2 * gd + 1)

    def _scale(self) -> float:
        raise NotImplementedError

    def _k_rank(self) -> int:
        return 0

    def _launch_device(self, ctx, d_X_ptr, d_thr_ptr, d_noise_ptr, d_mask_ptr, n_frames, R, D):
        """Run on device-resident planes (used by the range-Doppler detectors and the batch pipeline)."""
        (tr, td), (gr, gd) = self.num_train, self.num_guard
        _lib.check(ctx.lib.mmw_cfar2d(ctx.handle, d_X_ptr, d_thr_ptr, d_noise_ptr, d_mask_ptr, n_frames, R, D,
                                      self.kind, int(tr), int(td), int(gr), int(gd), float(self._scale()),
                                      int(self._k_rank())))

    def _compute_thresholds(self, X: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        X = np.asarray(X)
        R, D = X.shape

        def launch(ctx, d_x, d_t, d_n, d_m):
            self._launch_device(ctx, d_x.ptr, d_t.ptr, d_n.ptr, d_m.ptr, 1, R, D)
        thr, noise, self._mask = self._run(X, launch)
        return thr, noise

    def detect(self, X: np.ndarray) -> List[Tuple[int, int]]:
        X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("Input X must be a 2D array.")
        self.thresholds, self.noise_estimates = self._compute_thresholds(X)      # the subclass hook (reference :222-226)
        self.detections = X > self.thresholds
        rows, cols = np.where(self.detections)
        return list(zip(rows, cols))

    def _get_window_view(self, X: np.ndarray) -> np.ndarray:
        """2-D sliding windows (reference base.py:308-327), for subclasses with their own host-side thresholds."""
        win_r = 2 * (self.num_train[0] + self.num_guard[0]) + 1
        win_d = 2 * (self.num_train[1] + self.num_guard[1]) + 1
        if X.shape[0] < win_r or X.shape[1] < win_d:
            raise ValueError(f"Input shape {X.shape} is smaller than window size {(win_r, win_d)}.")
        return sliding_window_view(X, window_shape=(win_r, win_d))

    def plot_detections(self, X, title="2D CFAR Detection"):  # pragma: no cover
        raise NotImplementedError("plotting is outside the accelerated hot path (SURVEY.md section 2)")
