"""String keys -> range-Doppler detector classes (the keys of the reference's
mmwave_radar_processing/processors/range_doppler_detection/registry.py:10-23).

A detector class announces its key with ``@rd_detector("key")`` where it is defined; importing the package registers the
three shipped ones, and a user class can join the same table (``PointCloudGenerator(detector_type=...)`` looks keys up here).
"""
_BY_KEY = {}


def rd_detector(key: str):
    def register(cls):
        _BY_KEY[key] = cls
        return cls
    return register


def get_range_doppler_detector_registry():
    from . import range_doppler_detector_2d, range_doppler_detector_sequential, range_doppler_ground_detector  # noqa: F401
    return dict(_BY_KEY)
