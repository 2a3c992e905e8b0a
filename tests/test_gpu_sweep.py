"""Many-frame parity sweep: CFAR detection indices of the device-resident pipeline vs the oracle, bit-exact.

Default 96 frames (about 6 s of oracle time on one core); set MMW_SWEEP_FRAMES / MMW_SWEEP_PROCS for the long
run quoted in DESIGN.md (3000 frames, 16 processes).  Frames come from the device-side generator, so the
oracle sees exactly the bytes the GPU processed."""
import os
from multiprocessing import get_context

import numpy as np
import pytest

from mmwave_radar_processing_amd import synth
from mmwave_radar_processing_amd.batch import FramePipeline
from mmwave_radar_processing_amd.config_managers import ConfigManager
from mmwave_radar_processing_amd.detectors import CaCFAR2D

pytestmark = pytest.mark.gpu


def _oracle_dets(cube):
    from oracle import oracle_np as O
    return O.rd_detect_2d(cube)[2]


# the headline cube (fused register-resident RD kernel) and the shape of the 6843 ods cfgs the reference ships
# (63 samples x 100 loops: mixed-radix LDS kernel, float64 CFAR plane through k_rd_mixed<double>)
@pytest.mark.parametrize("shape,scale", [((12, 256, 128), 1), ((12, 63, 100), 4)])
def test_detection_indices_bit_exact_over_many_frames(shape, scale):
    n_frames = int(os.environ.get("MMW_SWEEP_FRAMES", "96")) * scale
    procs = int(os.environ.get("MMW_SWEEP_PROCS", "4"))
    cm = ConfigManager()
    cm.load_cfg_text(synth.synth_cfg_text(num_samples=shape[1], num_loops=shape[2]))
    batch = min(n_frames, 256 * scale)
    pipe = FramePipeline(cm, max_frames=batch, shape=shape, cfar=CaCFAR2D((4, 4), (2, 2), 1e-5))
    total_dets = mismatched = 0
    with get_context("spawn").Pool(procs) as pool:
        for f0 in range(0, n_frames, batch):
            nf = min(batch, n_frames - f0)
            pipe.synth(nf, seed0=900_000 + f0)
            dets = pipe.detect()
            cubes = pipe.cubes(0, nf)
            ref = pool.map(_oracle_dets, [cubes[i] for i in range(nf)], chunksize=4)
            for f in range(nf):
                total_dets += ref[f].shape[0]
                if not np.array_equal(dets[f], ref[f]):
                    mismatched += 1
    print(f"sweep {shape}: {n_frames} frames, {total_dets} detections, {mismatched} frames with any index difference")
    assert mismatched == 0
    assert total_dets > 5 * n_frames
