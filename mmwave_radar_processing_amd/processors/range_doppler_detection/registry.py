"""String keys -> range-Doppler detector classes
(reference: mmwave_radar_processing/processors/range_doppler_detection/registry.py:10-23).

``range_doppler_ground_detector`` is registered by the reference too, but it is built on the stateful
Altimeter (last-altitude gate, scipy ZoomFFT) which SURVEY.md section 8(e) keeps out of the shardable hot
path; asking for it raises a ValueError that says so instead of silently substituting another detector."""
from .range_doppler_detector_2d import RangeDopplerDetector2D
from .range_doppler_detector_sequential import RangeDopplerDetectorSequential

_REGISTRY = {
    "range_doppler_detector_2d": RangeDopplerDetector2D,
    "range_doppler_detector_sequential": RangeDopplerDetectorSequential,
}


def get_range_doppler_detector_registry():
    return dict(_REGISTRY)
