"""Cell-averaging CFAR (reference: mmwave_radar_processing/detectors/ca_cfar.py:11-155)."""
from .. import _lib
from .base import BaseCFAR1D, BaseCFAR2D, compute_alpha_ca


class CaCFAR1D(BaseCFAR1D):
    """T = alpha * mean(2*num_train training cells), alpha from Pfa (ca_cfar.py:43-60)."""
    kind = _lib.CFAR_CA

    def _scale(self):
        return compute_alpha_ca(2 * self.num_train, self.pfa)


class CaCFAR2D(BaseCFAR2D):
    """T = alpha * mean(training ring), N = window cells minus guard block (ca_cfar.py:111-140)."""
    kind = _lib.CFAR_CA

    def _scale(self):
        return compute_alpha_ca(self.num_train_cells(), self.pfa)
