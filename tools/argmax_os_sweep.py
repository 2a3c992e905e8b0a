#!/usr/bin/env python3
"""The stand-alone exact argmax (mmw_angle_argmax_exact through FramePipeline.point_clouds) on OS-CFAR detections -- ~470 mostly
noise-level cells per 256 x 128 frame, the hard case for the certainty test -- against the float64 oracle.
    python tools/argmax_os_sweep.py            # default bound (1/8 + pairwise pass)
    MMW_ARGMAX_BOUND_DIV=1 python tools/argmax_os_sweep.py"""
import sys, os, numpy as np, tempfile
sys.path.insert(0, os.getcwd())
from oracle import oracle_np as O
from mmwave_radar_processing_amd import _lib, synth
from mmwave_radar_processing_amd.batch import FramePipeline
from mmwave_radar_processing_amd.detectors import OsCFAR2D
from mmwave_radar_processing_amd.config_managers.cfgManager import ConfigManager
def make_cm(text):
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
        f.write(text); p = f.name
    cm = ConfigManager(); cm.load_cfg(p); cm.compute_radar_perforance(profile_idx=0); return cm
cm = make_cm(synth.SYNTH_CFG_256x128x12)
F = 160
cubes = np.stack([synth.synth_cube(880000 + f) for f in range(F)])
az, el = list(range(8)), [8, 9, 10, 11]
pipe = FramePipeline(cm, max_frames=F, shape=(12, 256, 128), cfar=OsCFAR2D((5, 5), (3, 2), rho=0.7, alpha=2.0),
                     az_antenna_idxs=az, el_antenna_idxs=el, det_capacity=2048)
pipe.load(cubes)
assert not pipe._fused_supported(True)
pipe.point_clouds()
bad_az = bad_el = n = 0
for f in range(F):
    raw = O.range_doppler(cubes[f])
    d = pipe.dets[f]
    r, v = d[:, 0].astype(int), d[:, 1].astype(int)
    bad_az += int((pipe.az_idx[f] != O.angle_argmax(raw, r, v, az, 64, True)[0]).sum())
    bad_el += int((pipe.el_idx[f] != O.angle_argmax(raw, r, v, el, 64, False)[0]).sum())
    n += len(r)
print(f"stand-alone exact argmax (bound divisor {os.environ.get('MMW_ARGMAX_BOUND_DIV', '8')} + pairwise pass), OS-CFAR detections: {F} frames, {n} detections, "
      f"{bad_az} azimuth / {bad_el} elevation index differences vs the float64 oracle, {pipe.n_refined} evaluations refined")
