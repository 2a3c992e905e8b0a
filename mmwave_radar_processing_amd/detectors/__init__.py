"""CFAR detectors computed on the GPU; class names and registry keys follow the reference's detectors package."""
from . import base, ca_cfar, go_so_cfar, os_cfar
from .detector_registry import get_detector_registry

BaseCFAR1D, BaseCFAR2D = base.BaseCFAR1D, base.BaseCFAR2D
# every registered detector class is importable from the package: CaCFAR1D, CaCFAR2D, OsCFAR1D, ...
globals().update({cls.__name__: cls for cls in get_detector_registry().values()})

__all__ = ["BaseCFAR1D", "BaseCFAR2D", "get_detector_registry"] + \
    sorted(cls.__name__ for cls in get_detector_registry().values())
