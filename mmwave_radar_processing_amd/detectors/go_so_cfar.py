"""Greatest-of / smallest-of CFAR (reference: mmwave_radar_processing/detectors/go_so_cfar.py:11-123).

alpha uses ONE side's cell count, as the reference does (:56-59,111)."""
from .. import _lib
from .base import BaseCFAR1D, compute_alpha_ca


class GoCFAR1D(BaseCFAR1D):
    kind = _lib.CFAR_GO

    def _scale(self):
        return compute_alpha_ca(self.num_train, self.pfa)


class SoCFAR1D(BaseCFAR1D):
    kind = _lib.CFAR_SO

    def _scale(self):
        return compute_alpha_ca(self.num_train, self.pfa)
