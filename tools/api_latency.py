#!/usr/bin/env python3
"""Per-call latency of the drop-in Python API on one frame (host cube in, NumPy results out).
This path is bound by PCIe copies and dtype conversions (complex64 -> complex128), not by HBM."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import synth  # noqa: E402
from mmwave_radar_processing_amd.config_managers import ConfigManager  # noqa: E402
from mmwave_radar_processing_amd.processors import (PointCloudGenerator, RangeAngleProcessorDBSEnhanced,  # noqa: E402
                                                    RangeDopplerProcessor)
from mmwave_radar_processing_amd.processors.range_doppler_detection import RangeDopplerDetector2D  # noqa: E402


def timeit(fn, n=30):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return 1e3 * (time.perf_counter() - t0) / n


cm = ConfigManager()
cm.load_cfg_text(synth.SYNTH_CFG_256x128x12)
cube = synth.synth_cube(3).astype(np.complex128)
cfar = {"cfar_type": "ca_cfar_2d", "cfar_params": {"num_train": (4, 4), "num_guard": (2, 2), "pfa": 1e-5}}
rd = RangeDopplerProcessor(cm)
det = RangeDopplerDetector2D(cm, **cfar)
pcg = PointCloudGenerator(cm, az_antenna_idxs=list(range(8)), el_antenna_idxs=[8, 9, 10, 11], detector_params=cfar)
dbs = RangeAngleProcessorDBSEnhanced(cm)
out = {
    "RangeDopplerProcessor.process(rx_idx=0) ms": timeit(lambda: rd.process(cube, rx_idx=0)),
    "RangeDopplerProcessor.process(rx_idx=-1, complex) ms": timeit(lambda: rd.process(cube, rx_idx=-1, return_magnitude=False)),
    "RangeDopplerDetector2D.process ms": timeit(lambda: det.process(cube)),
    "PointCloudGenerator.process ms": timeit(lambda: pcg.process(cube)),
    "compute_3d_windowed_fft ms": timeit(lambda: dbs.compute_3d_windowed_fft(cube), 10),
}
print(json.dumps({k: round(v, 3) for k, v in out.items()}, indent=1))
