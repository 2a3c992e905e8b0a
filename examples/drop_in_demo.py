#!/usr/bin/env python3
"""Drop-in usage: the reference's processor protocol, computed on an MI355X.

    python examples/drop_in_demo.py

Mirrors what the reference's plugin host does per frame
(mmwave_radar_processing/visualization/backends/view_controller.py:94-111): construct processors from a
ConfigManager and YAML-style parameter dicts, call process(adc_cube=..., **params), read cached attributes.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import synth  # noqa: E402
from mmwave_radar_processing_amd.batch import FramePipeline  # noqa: E402
from mmwave_radar_processing_amd.config_managers import ConfigManager  # noqa: E402
from mmwave_radar_processing_amd.detectors import CaCFAR2D  # noqa: E402
from mmwave_radar_processing_amd.processors import (PointCloudGenerator, RangeAngleProcessorDBSEnhanced,  # noqa: E402
                                                    RangeDopplerProcessor, VirtualArrayReformatter)

cm = ConfigManager()
cm.load_cfg_text(synth.SYNTH_CFG_256x128x12)
raw = synth.synth_raw_cube(seed=1)                              # [4 Rx, 256 samples, 3 Tx * 128 loops]
cube = VirtualArrayReformatter(cm).process(raw)                  # [12, 256, 128] complex128

params = {"rx_idx": 0}                                          # gui_configs/processor_params.yaml style
rd = RangeDopplerProcessor(cm, **params).process(adc_cube=cube, **params)
print("range-Doppler magnitude", rd.shape, rd.dtype, "peak at", np.unravel_index(np.argmax(rd), rd.shape))

pcg = PointCloudGenerator(cm, az_antenna_idxs=list(range(8)), el_antenna_idxs=[8, 9, 10, 11],
                          detector_type="range_doppler_detector_2d",
                          detector_params={"cfar_type": "ca_cfar_2d",
                                           "cfar_params": {"num_train": [4, 4], "num_guard": [2, 2], "pfa": 1e-5}})
points = pcg.process(cube)
print("point cloud", points.shape, "| detections", pcg.detector.dets.shape, "| thresholds", pcg.detector.detector.thresholds.shape)

cube3d = RangeAngleProcessorDBSEnhanced(cm).compute_3d_windowed_fft(cube)
print("angle-range-Doppler cube", cube3d.shape, cube3d.dtype)

# batch: 64 frames generated in HBM, detections + point clouds for all of them in one pass
pipe = FramePipeline(cm, max_frames=64, shape=(12, 256, 128), cfar=CaCFAR2D((4, 4), (2, 2), 1e-5),
                     az_antenna_idxs=range(8), el_antenna_idxs=[8, 9, 10, 11])
pipe.synth(64, seed0=2024)
clouds = pipe.point_clouds()
print("batch:", len(clouds), "frames,", sum(c.shape[0] for c in clouds), "points")
