"""Ordered-statistic CFAR (reference: mmwave_radar_processing/detectors/os_cfar.py:11-195).

``rho`` is a fraction of the training-cell count; k = clamp(int(rho*N), 1, N)
(:25-27,131-132).  ``pfa`` is unused (set to 0.0 like the reference)."""
from .. import _lib
from .base import BaseCFAR1D, BaseCFAR2D


def _rank(rho, n_cells):
    return max(1, min(int(rho * n_cells), n_cells))


class OsCFAR1D(BaseCFAR1D):
    kind = _lib.CFAR_OS

    def __init__(self, num_train, num_guard, rho, alpha, **kwargs):
        super().__init__(num_train, num_guard, pfa=0.0, **kwargs)
        self.alpha = alpha
        self.k_rank = _rank(rho, 2 * num_train)

    def _scale(self):
        return self.alpha

    def _k_rank(self):
        return self.k_rank


class OsCFAR2D(BaseCFAR2D):
    kind = _lib.CFAR_OS

    def __init__(self, num_train, num_guard, rho, alpha, **kwargs):
        super().__init__(num_train, num_guard, pfa=0.0, **kwargs)
        self.alpha = alpha
        self.k_rank = _rank(rho, self.num_train_cells())

    def _scale(self):
        return self.alpha

    def _k_rank(self):
        return self.k_rank
