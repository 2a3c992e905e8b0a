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
from mmwave_radar_processing_amd.detectors import CaCFAR2D, OsCFAR2D

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


def _oracle_dets_os(cube):
    from oracle import oracle_np as O
    mag = np.abs(O.range_doppler(cube)[0])
    return np.array(O.os_cfar_2d(mag, (5, 5), (3, 2), 0.75, 3.0)[2], dtype=np.int64).reshape(-1, 2)


@pytest.mark.parametrize("shape", [(12, 256, 128), (12, 63, 100)])
def test_os_cfar_detection_indices_bit_exact_over_many_frames(shape):
    """Same sweep with the detector the reference's GUI config uses (os_cfar_2d, (5,5)/(3,2) window,
    gui_configs/processor_params.yaml): register-sorted tile ranks + bucket walk against np.partition."""
    n_frames = max(8, int(os.environ.get("MMW_SWEEP_FRAMES", "96")) // 4)
    procs = int(os.environ.get("MMW_SWEEP_PROCS", "4"))
    cm = ConfigManager()
    cm.load_cfg_text(synth.synth_cfg_text(num_samples=shape[1], num_loops=shape[2]))
    pipe = FramePipeline(cm, max_frames=n_frames, shape=shape, cfar=OsCFAR2D((5, 5), (3, 2), rho=0.75, alpha=3.0),
                         det_capacity=8192)
    pipe.synth(n_frames, seed0=700_000)
    dets = pipe.detect()
    cubes = pipe.cubes(0, n_frames)
    with get_context("spawn").Pool(procs) as pool:
        ref = pool.map(_oracle_dets_os, [cubes[i] for i in range(n_frames)], chunksize=2)
    total = sum(r.shape[0] for r in ref)
    bad = sum(0 if np.array_equal(dets[f], ref[f]) else 1 for f in range(n_frames))
    print(f"OS sweep {shape}: {n_frames} frames, {total} detections, {bad} frames with any index difference")
    assert bad == 0 and total > 0
