"""String keys -> range-Doppler detector classes
(reference: mmwave_radar_processing/processors/range_doppler_detection/registry.py:10-23)."""
from .range_doppler_detector_2d import RangeDopplerDetector2D
from .range_doppler_detector_sequential import RangeDopplerDetectorSequential
from .range_doppler_ground_detector import RangeDopplerGroundDetector

_REGISTRY = {
    "range_doppler_detector_2d": RangeDopplerDetector2D,
    "range_doppler_detector_sequential": RangeDopplerDetectorSequential,
    "range_doppler_ground_detector": RangeDopplerGroundDetector,        # stateful (Altimeter): single-frame API only
}


def get_range_doppler_detector_registry():
    return dict(_REGISTRY)
