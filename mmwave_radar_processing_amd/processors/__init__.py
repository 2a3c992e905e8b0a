from ._processor import _Processor
from .virtual_array_reformater import VirtualArrayReformatter
from .range_resp import RangeProcessor
from .altimeter import Altimeter
from .range_doppler_resp import RangeDopplerProcessor
from .range_angle_resp import RangeAngleProcessor
from .range_angle_resp_dbs_enhanced import RangeAngleProcessorDBSEnhanced
from .point_cloud_generator import PointCloudGenerator
from .doppler_azimuth_resp import DopplerAzimuthProcessor

__all__ = ["_Processor", "VirtualArrayReformatter", "RangeProcessor", "Altimeter", "RangeDopplerProcessor",
           "RangeAngleProcessor", "RangeAngleProcessorDBSEnhanced", "PointCloudGenerator", "DopplerAzimuthProcessor"]
