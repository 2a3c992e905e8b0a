"""String keys -> detector classes (reference: mmwave_radar_processing/detectors/detector_registry.py:15-27)."""
from .ca_cfar import CaCFAR1D, CaCFAR2D
from .go_so_cfar import GoCFAR1D, SoCFAR1D
from .os_cfar import OsCFAR1D, OsCFAR2D

_REGISTRY = {
    "ca_cfar_1d": CaCFAR1D, "ca_cfar_2d": CaCFAR2D,
    "os_cfar_1d": OsCFAR1D, "os_cfar_2d": OsCFAR2D,
    "go_cfar_1d": GoCFAR1D, "so_cfar_1d": SoCFAR1D,
}


def get_detector_registry():
    return dict(_REGISTRY)


def make_detector(cfar_type: str, cfar_params=None):
    """Instance for a registry key; an unknown key is the ValueError every caller of the reference's registry raises."""
    cls = _REGISTRY.get(cfar_type)
    if cls is None:
        raise ValueError(f"Unknown CFAR type: {cfar_type}. Available: {list(_REGISTRY)}")
    return cls(**(cfar_params or {}))
