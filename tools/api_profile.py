#!/usr/bin/env python3
"""cProfile of the single-frame detector / point-cloud calls of the drop-in API (where do the milliseconds go?)."""
import cProfile
import os
import pstats
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import synth  # noqa: E402
from mmwave_radar_processing_amd.config_managers import ConfigManager  # noqa: E402
from mmwave_radar_processing_amd.processors import PointCloudGenerator  # noqa: E402
from mmwave_radar_processing_amd.processors.range_doppler_detection import RangeDopplerDetector2D  # noqa: E402

cm = ConfigManager()
cm.load_cfg_text(synth.SYNTH_CFG_256x128x12)
cube = synth.synth_cube(3).astype(np.complex128)
cfar = {"cfar_type": "ca_cfar_2d", "cfar_params": {"num_train": (4, 4), "num_guard": (2, 2), "pfa": 1e-5}}
det = RangeDopplerDetector2D(cm, **cfar)
pcg = PointCloudGenerator(cm, az_antenna_idxs=list(range(8)), el_antenna_idxs=[8, 9, 10, 11], detector_params=cfar)
for name, fn in (("detector", lambda: det.process(cube)), ("point cloud", lambda: pcg.process(cube))):
    fn()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        fn()
    pr.disable()
    print("=====", name)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
