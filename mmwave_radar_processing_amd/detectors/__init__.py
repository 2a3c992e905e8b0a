from .base import BaseCFAR1D, BaseCFAR2D
from .ca_cfar import CaCFAR1D, CaCFAR2D
from .go_so_cfar import GoCFAR1D, SoCFAR1D
from .os_cfar import OsCFAR1D, OsCFAR2D
from .detector_registry import get_detector_registry

__all__ = ["BaseCFAR1D", "BaseCFAR2D", "CaCFAR1D", "CaCFAR2D", "GoCFAR1D", "SoCFAR1D",
           "OsCFAR1D", "OsCFAR2D", "get_detector_registry"]
