from .range_doppler_detector import RangeDopplerDetector
from .range_doppler_detector_2d import RangeDopplerDetector2D
from .range_doppler_detector_sequential import RangeDopplerDetectorSequential
from .range_doppler_ground_detector import RangeDopplerGroundDetector
from .registry import get_range_doppler_detector_registry

__all__ = ["RangeDopplerDetector", "RangeDopplerDetector2D", "RangeDopplerDetectorSequential", "RangeDopplerGroundDetector",
           "get_range_doppler_detector_registry"]
