#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the IMPORTED reference.

Run in the build container only (the reference never travels):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference:/root/repo \
        python tests/golden/make_golden.py

Inputs are regenerated from seeds (``mmwave_radar_processing_amd.synth`` /
``np.random.seed(42)``), so only the reference's OUTPUTS are stored, plus the
command lines of the reference's ``configs/*.cfg`` data files (TI-format
data, needed because ``/root/reference`` does not exist on the GPU box).
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from mmwave_radar_processing.config_managers.cfgManager import ConfigManager            # noqa: E402
from mmwave_radar_processing.processors.virtual_array_reformater import VirtualArrayReformatter  # noqa: E402
from mmwave_radar_processing.processors.range_resp import RangeProcessor                # noqa: E402
from mmwave_radar_processing.processors.range_doppler_resp import RangeDopplerProcessor  # noqa: E402
from mmwave_radar_processing.processors.range_angle_resp import RangeAngleProcessor     # noqa: E402
from mmwave_radar_processing.processors.range_angle_resp_dbs_enhanced import RangeAngleProcessorDBSEnhanced  # noqa: E402
from mmwave_radar_processing.processors.range_doppler_detection.range_doppler_detector_2d import RangeDopplerDetector2D  # noqa: E402
from mmwave_radar_processing.processors.range_doppler_detection.range_doppler_detector_sequential import RangeDopplerDetectorSequential  # noqa: E402
from mmwave_radar_processing.processors.range_doppler_detection.range_doppler_ground_detector import RangeDopplerGroundDetector  # noqa: E402
from mmwave_radar_processing.processors.point_cloud_generator import PointCloudGenerator  # noqa: E402
from mmwave_radar_processing.processors.simple_synthetic_array_beamformer_processor_multiFrame import SyntheticArrayBeamformerProcessor  # noqa: E402
from mmwave_radar_processing.processors.doppler_azimuth_resp import DopplerAzimuthProcessor      # noqa: E402
from mmwave_radar_processing.detectors import CaCFAR1D, CaCFAR2D, GoCFAR1D, SoCFAR1D, OsCFAR1D, OsCFAR2D  # noqa: E402

from mmwave_radar_processing_amd import synth                                          # noqa: E402

CFG_KEYS = ("channelCfg", "adcCfg", "adcbufCfg", "profileCfg", "chirpCfg", "frameCfg")


def load_cm(text):
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
        f.write(text)
        path = f.name
    cm = ConfigManager()
    cm.load_cfg(path)
    os.unlink(path)
    return cm


def cm_scalars(cm):
    return dict(
        num_rx=cm.num_rx_antennas, num_tx=cm.num_tx_antennas,
        num_samples=cm.get_num_adc_samples(0), loops=cm.frameCfg_loops,
        frame_start=cm.frameCfg_start_index, frame_end=cm.frameCfg_end_index,
        range_res_m=cm.range_res_m, range_max_m=cm.range_max_m,
        range_bin_size_m=float(cm.range_bin_size_m),
        vel_res_m_s=cm.vel_res_m_s, vel_max_m_s=cm.vel_max_m_s,
        virtual_antennas_enabled=bool(cm.virtual_antennas_enabled),
    )


def gen_cfgs():
    out = {}
    cdir = os.path.join(REF, "configs")
    for name in sorted(os.listdir(cdir)):
        if not name.endswith(".cfg"):
            continue
        with open(os.path.join(cdir, name)) as f:
            lines = [ln.rstrip("\n") for ln in f if ln.split(" ")[0].strip() in CFG_KEYS and "%" not in ln]
        cm = ConfigManager()
        cm.load_cfg(os.path.join(cdir, name))
        out[name] = dict(lines=lines, expect=cm_scalars(cm))
    out["__synth_256x128x12__"] = dict(lines=synth.SYNTH_CFG_256x128x12.splitlines(),
                                       expect=cm_scalars(load_cm(synth.SYNTH_CFG_256x128x12)))
    with open(os.path.join(HERE, "cfg_scalars.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("cfg_scalars.json:", len(out), "configs")


def checksum(a):
    a = np.asarray(a)
    return np.array([np.sum(a), np.sum(np.abs(a)), np.max(np.abs(a))], dtype=np.complex128)


def gen_small_chain():
    """Full outputs on a small power-of-two cube and subsampled outputs on the sample cfg's (12,63,70)."""
    d = {}
    # --- (12, 32, 16)
    txt = synth.synth_cfg_text(num_samples=32, num_loops=16)
    cm = load_cm(txt)
    cube = synth.synth_cube(101, (12, 32, 16))
    rdp = RangeDopplerProcessor(cm)
    d["p2_rd"] = rdp.process(cube, rx_idx=-1, return_magnitude=False)
    d["p2_rd_mag_rx3"] = rdp.process(cube, rx_idx=3, return_magnitude=True)
    d["p2_range_bins"], d["p2_vel_bins"] = rdp.range_bins, rdp.vel_bins
    rp = RangeProcessor(cm)
    d["p2_range_profile_c5"] = rp.process(cube, chirp_idx=5)
    rap = RangeAngleProcessor(cm, num_angle_bins=64)
    d["p2_ra_all"] = rap.process(cube, chirp_idx=2)
    d["p2_ra_sub"] = rap.process(cube, chirp_idx=0, rx_antennas=[0, 3, 4, 7])
    d["p2_ra_nowin"] = rap.process(cube, chirp_idx=1, rx_antennas=[1, 2], perform_windowing=False)
    d["p2_ra_range_bins"], d["p2_angle_bins"], d["p2_phase_shifts"] = rap.range_bins, rap.angle_bins, rap.phase_shifts
    dbs = RangeAngleProcessorDBSEnhanced(cm, num_angle_bins_range_angle_response=64,
                                         num_angle_bins_dbs_enhanced_response=40)
    f3 = dbs.compute_3d_windowed_fft(cube)
    d["p2_fft3d"] = f3
    vel = np.array([0.6, -0.2, 0.05])
    d["p2_dbs_vel"] = vel
    d["p2_dbs"] = dbs.process(cube, velocity_ned=vel)
    d["p2_dbs_angle_bins"] = dbs.angle_bins_dbs_enhanced
    # --- sample cfg shape (12, 63, 70): non power of two
    with open(os.path.join(REF, "configs", "6843_RadVel_ods_20Hz.cfg")) as f:
        cm2 = load_cm(f.read())
    raw = synth.synth_raw_cube(202, 4, 3, 63, 70)
    var = VirtualArrayReformatter(cm2)
    virt = var.process(raw)
    d["np2_virt_checksum"] = checksum(virt)
    d["np2_virt_sample"] = virt[:, ::7, ::9]
    rdp2 = RangeDopplerProcessor(cm2)
    rd2 = rdp2.process(virt, rx_idx=-1, return_magnitude=False)
    d["np2_rd_sample"] = rd2[:, ::3, ::5]
    d["np2_rd_checksum"] = checksum(rd2)
    det2 = RangeDopplerDetector2D(cm2, cfar_type="ca_cfar_2d",
                                  cfar_params={"num_train": (4, 4), "num_guard": (2, 2), "pfa": 1e-5})
    d["np2_dets"] = det2.process(virt)
    d["np2_mag0"] = det2.rng_dop_resp
    dbs2 = RangeAngleProcessorDBSEnhanced(cm2)
    f32 = dbs2.compute_3d_windowed_fft(virt)
    d["np2_fft3d_sample"] = f32[::4, ::3, ::5]
    d["np2_fft3d_checksum"] = checksum(f32)
    # small reformatter case, full output
    cm3 = load_cm(synth.synth_cfg_text(num_samples=16, num_loops=8))
    raw3 = synth.synth_raw_cube(303, 4, 3, 16, 8)
    d["var_small"] = VirtualArrayReformatter(cm3).process(raw3)
    np.savez_compressed(os.path.join(HERE, "small_chain.npz"), **d)
    print("small_chain.npz:", sum(v.nbytes for v in d.values()) // 1024, "KiB raw")


def gen_frames_256():
    """Headline shape (12,256,128): detections, point clouds, subsampled spectra, checksums."""
    cm = load_cm(synth.SYNTH_CFG_256x128x12)
    d = {}
    cfar = {"num_train": (4, 4), "num_guard": (2, 2), "pfa": 1e-5}
    pcg = PointCloudGenerator(cm, az_antenna_idxs=list(range(8)), el_antenna_idxs=[8, 9, 10, 11],
                              detector_type="range_doppler_detector_2d",
                              detector_params={"cfar_type": "ca_cfar_2d", "cfar_params": cfar},
                              num_angle_bins=64)
    dbs = RangeAngleProcessorDBSEnhanced(cm)
    d["angle_bins"] = pcg.angle_bins
    d["range_bins"], d["vel_bins"] = pcg.detector.range_bins, pcg.detector.vel_bins
    seeds = [0, 1, 2, 3]
    d["seeds"] = np.array(seeds)
    for s in seeds:
        cube = synth.synth_cube(s)
        pc = pcg.process(cube)
        det = pcg.detector
        d[f"s{s}_dets"] = det.dets
        d[f"s{s}_pc"] = pc
        raw = det.rng_dop_resp_raw
        if det.dets.shape[0]:
            az, el = pcg._compute_angle_estimation(raw, det.dets[:, 0], det.dets[:, 1])
            d[f"s{s}_az"], d[f"s{s}_el"] = az, el
        d[f"s{s}_rd_sample"] = raw[:, ::8, ::8]
        d[f"s{s}_rd_checksum"] = checksum(raw)
        thr = det.detector.thresholds
        d[f"s{s}_thr_sample"] = thr[::4, ::4]
        d[f"s{s}_noise_sample"] = det.detector.noise_estimates[::4, ::4]
        if s < 2:
            d[f"s{s}_mag0"] = det.rng_dop_resp
        f3 = dbs.compute_3d_windowed_fft(cube)
        d[f"s{s}_fft3d_sample"] = f3[::4, ::8, ::8]
        d[f"s{s}_fft3d_checksum"] = checksum(f3)
    # pure-noise frame: false-alarm sanity
    noise = synth.synth_cube(77, num_targets=0)
    det2 = RangeDopplerDetector2D(cm, cfar_type="ca_cfar_2d", cfar_params=cfar)
    d["noise77_dets"] = det2.process(noise)
    np.savez_compressed(os.path.join(HERE, "frames_256.npz"), **d)
    print("frames_256.npz:", sum(v.nbytes for v in d.values()) // 1024, "KiB raw;",
          [int(d[f's{s}_dets'].shape[0]) for s in seeds], "dets")


def gen_cfar_known():
    """Replay of tests/verify_detectors_manual.py:15-89 (np.random.seed(42)), outputs only."""
    d = {}
    np.random.seed(42)
    x = np.random.exponential(scale=1.0, size=100)
    x[50] = 10.0
    for name, det in (("ca", CaCFAR1D(10, 2, 1e-3)), ("go", GoCFAR1D(10, 2, 1e-3)),
                      ("so", SoCFAR1D(10, 2, 1e-3)),
                      ("os", OsCFAR1D(10, 2, rho=0.75, alpha=5.0)),
                      ("os_ascalled", OsCFAR1D(10, 2, 15, 5.0))):
        dets = det.detect(x)
        d[f"1d_{name}_dets"] = np.array(dets, dtype=np.int64)
        d[f"1d_{name}_thr"] = det.thresholds
        d[f"1d_{name}_noise"] = det.noise_estimates
    X = np.random.exponential(scale=1.0, size=(50, 50))
    X[25, 25] = 15.0
    for name, det in (("ca", CaCFAR2D((5, 5), (2, 2), 1e-4)),
                      ("os", OsCFAR2D((5, 5), (2, 2), rho=0.8, alpha=5.0)),
                      ("os_yaml", OsCFAR2D([5, 5], [3, 2], rho=0.7, alpha=2))):
        dets = det.detect(X)
        d[f"2d_{name}_dets"] = np.array(dets, dtype=np.int64).reshape(-1, 2)
        d[f"2d_{name}_thr"] = det.thresholds
        d[f"2d_{name}_noise"] = det.noise_estimates
    d["alpha_20_1e-3"] = np.array(CaCFAR1D.compute_alpha_ca(20, 1e-3))
    d["alpha_200_1e-4"] = np.array(CaCFAR2D.compute_alpha_ca(200, 1e-4))
    # too-small inputs -> all-inf, no error
    small = CaCFAR2D((4, 4), (2, 2), 1e-5)
    d["2d_small_dets_len"] = np.array(len(small.detect(np.ones((5, 5)))))
    np.savez_compressed(os.path.join(HERE, "cfar_known.npz"), **d)
    print("cfar_known.npz written; 1-D CA dets", d["1d_ca_dets"], "2-D CA dets", d["2d_ca_dets"].tolist())


def gen_bartlett():
    """Delay-and-sum contraction via the reference's compute_synthetic_response, driven on a bare namespace."""
    rng = np.random.default_rng(404)
    frames, S, chirps = 2, 64, 24
    hist = (rng.standard_normal((frames, S, chirps)) + 1j * rng.standard_normal((frames, S, chirps)))
    geom = rng.uniform(-0.02, 0.02, (frames, 3, chirps))
    az = np.linspace(-1.0, 1.0, 9)
    el = np.linspace(-0.3, 0.3, 3)
    ns = types.SimpleNamespace(history_acd_cube_valid_chirps=hist, az_angle_bins_rad=az,
                               el_angle_bins_rad=el, lambda_m=299792458.0 / 77e9,
                               num_range_bins=S, range_bins=np.arange(S))
    cls = SyntheticArrayBeamformerProcessor
    cls._compute_beam_stearing_vectors(ns)
    cls._init_out_resp(ns)
    ns.compute_response_at_steering_angle = types.MethodType(cls.compute_response_at_steering_angle, ns)
    out = cls.compute_synthetic_response(ns, geom)
    np.savez_compressed(os.path.join(HERE, "bartlett_small.npz"), out=out, d=ns.d, az=az, el=el,
                        lambda_m=np.array(ns.lambda_m))
    print("bartlett_small.npz:", out.shape)


def gen_doppler_azimuth():
    """DopplerAzimuthProcessor coarse path on a small cube (standard geometry) and the sample cfg shape (ods)."""
    d = {}
    cm = load_cm(synth.synth_cfg_text(num_samples=32, num_loops=16))
    cube = synth.synth_cube(101, (12, 32, 16))
    p = DopplerAzimuthProcessor(cm, num_angle_bins=64)
    d["std_all"] = p.process(cube)
    d["std_sub_win"] = p.process(cube, rx_antennas=[4, 5, 8, 9], range_window=[0.9, 2.0], shift_angle=False)
    d["valid_angle_bins"] = p.valid_angle_bins
    with open(os.path.join(REF, "configs", "6843_RadVel_ods_20Hz.cfg")) as f:
        text = f.read()
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
        f.write(text)
        path = f.name
    cm2 = ConfigManager()
    cm2.load_cfg(path, array_geometry="ods", array_direction="down")
    os.unlink(path)
    virt = synth.synth_cube(202, (12, 63, 70))
    p2 = DopplerAzimuthProcessor(cm2, num_angle_bins=64, valid_angle_range=np.array([-1.04719755, 1.04719755]))
    d["ods_sub"] = p2.process(virt, rx_antennas=[4, 5, 8, 9], range_window=[0.9, 2.0], shift_angle=False)
    # RangeProcessor.zoom_fft (scipy ZoomFFT inside the reference)
    rp = RangeProcessor(cm)
    z, zb = rp.zoom_fft(cube, range_start_m=0.6, range_stop_m=1.9, chirp_idx=3)
    d["zoom_mag"], d["zoom_bins"] = z, zb
    # precise (ZoomFFT) Doppler mode; the reference edits precise_vel_range in place, so every call gets a fresh array
    for tag, vr, kw in (("default", [-0.25, 0.25], {}),
                        ("pos_only", [0.3, 1.2], {}),
                        ("narrow", [-0.05, 0.02], {}),
                        ("clamped", [-50.0, 50.0], {"shift_angle": False}),
                        ("neg_sub", [-1.0, -0.2], {"rx_antennas": [4, 5, 8, 9], "range_window": [0.9, 2.0]})):
        d["precise_" + tag] = p.process(cube, use_precise_fft=True, precise_vel_range=np.array(vr), **kw)
        d["precise_" + tag + "_bins"] = np.array(p.zoomed_vel_bins)
    d["precise_ods"] = p2.process(virt, rx_antennas=[4, 5, 8, 9], range_window=[0.9, 2.0], use_precise_fft=True,
                                  precise_vel_range=np.array([-0.25, 0.25]))
    d["precise_ods_bins"] = np.array(p2.zoomed_vel_bins)
    # host-side scipy peak pickers that subclasses of these processors call (velocity_estimator.py, altimeter.py)
    d["peaks_rows_std_all"] = p.detect_peaks_rows(d["std_all"], p.vel_bins, 30.0)
    d["peaks_rows_precise"] = p.detect_peaks_rows(d["precise_default"], d["precise_default_bins"], 20.0)
    d["peak_zero_az_std_all"] = np.asarray(p.detect_peak_zero_az(d["std_all"], p.vel_bins, 30.0))
    d["peak_zero_az_ods"] = np.asarray(p2.detect_peak_zero_az(d["ods_sub"], p2.vel_bins, 30.0))
    prof = rp.process(cube, chirp_idx=2)
    d["range_profile_chirp2"] = prof
    pk_r, pk_v = rp.find_peaks(20 * np.log10(prof), rp.range_bins, max_peaks=3)
    d["range_peaks_m"], d["range_peaks_db"] = pk_r, pk_v
    np.savez_compressed(os.path.join(HERE, "doppler_azimuth.npz"), **d)
    print("doppler_azimuth.npz:", {k: v.shape for k, v in d.items()})


# detector parameters of the reference's shipped GUI config (gui_configs/processor_params.yaml:40-86)
YAML_OS2D = {"num_train": [5, 5], "num_guard": [3, 2], "rho": 0.7, "alpha": 2}
YAML_SEQ = dict(rng_cfar_type="os_cfar_1d", rng_cfar_params={"num_train": 5, "num_guard": 3, "rho": 0.6, "alpha": 2},
                vel_cfar_type="os_cfar_1d", vel_cfar_params={"num_train": 5, "num_guard": 2, "rho": 0.7, "alpha": 3})
GOSO_SEQ = dict(rng_cfar_type="go_cfar_1d", rng_cfar_params={"num_train": 8, "num_guard": 2, "pfa": 1e-3},
                vel_cfar_type="so_cfar_1d", vel_cfar_params={"num_train": 6, "num_guard": 2, "pfa": 1e-4})
YAML_GROUND = dict(vel_cfar_type="os_cfar_1d", vel_cfar_params={"num_train": 12, "num_guard": 4, "rho": 0.5, "alpha": 15},
                   altimeter_params={"min_altitude_m": 0.25, "zoom_search_region_m": 0.2, "altitude_search_limit_m": 0.4,
                                     "range_bias": 0.0, "precise_est_enabled": False})
COARSE_GROUND = dict(vel_cfar_type="os_cfar_1d", vel_cfar_params={"num_train": 12, "num_guard": 4, "rho": 0.5, "alpha": 6},
                     altimeter_params={"min_altitude_m": 0.6, "zoom_search_region_m": 0.2, "altitude_search_limit_m": 0.6,
                                       "range_bias": 0.0, "precise_est_enabled": False})
PRECISE_GROUND = dict(vel_cfar_type="os_cfar_1d", vel_cfar_params={"num_train": 16, "num_guard": 4, "rho": 0.5, "alpha": 12},
                      altimeter_params={"min_altitude_m": 0.6, "zoom_search_region_m": 0.2, "altitude_search_limit_m": 0.6,
                                        "range_bias": 0.03, "precise_est_enabled": True})


def gen_detectors_rd():
    """The detectors the reference's shipped YAMLs actually use, run by the reference on range-Doppler data:
    RangeDopplerDetectorSequential, RangeDopplerDetector2D with os_cfar_2d, RangeDopplerGroundDetector (stateful, over a
    5-frame sequence), PointCloudGenerator on the ground detector, and the 3-D chain on cubes with a non-finite sample."""
    d = {}
    cm = load_cm(synth.SYNTH_CFG_256x128x12)
    with open(os.path.join(REF, "configs", "6843_RadVel_ods_20Hz.cfg")) as f:
        cm2 = load_cm(f.read())
    cases = [(f"s{s}", cm, synth.synth_cube(s)) for s in (0, 1, 2, 3)] + [("np2", cm2, synth.synth_cube(202, (12, 63, 70)))]
    for tag, c, cube in cases:
        d[f"{tag}_seq_yaml"] = RangeDopplerDetectorSequential(c, **YAML_SEQ).process(cube)
        d[f"{tag}_seq_goso"] = RangeDopplerDetectorSequential(c, **GOSO_SEQ).process(cube)
        d[f"{tag}_os2d_yaml"] = RangeDopplerDetector2D(c, cfar_type="os_cfar_2d", cfar_params=YAML_OS2D).process(cube)
    # ground detector: stateful over the sequence; then reset() and the first two frames again
    seq = synth.synth_ground_sequence(606, 5)
    for name, params in (("yaml", YAML_GROUND), ("coarse", COARSE_GROUND), ("precise", PRECISE_GROUND)):
        det = RangeDopplerGroundDetector(cm, **params)
        alts = []
        for f in range(seq.shape[0]):
            d[f"ground_{name}_f{f}"] = det.process(seq[f])
            alts.append(det.altimeter.current_altitude_corrected_m)
        d[f"ground_{name}_alt"] = np.array(alts)
        det.reset()
        d[f"ground_{name}_after_reset_f3"] = det.process(seq[3])
        d[f"ground_{name}_after_reset_alt"] = np.array(det.altimeter.current_altitude_corrected_m)
    pcg = PointCloudGenerator(cm, az_antenna_idxs=[0, 3, 4, 7], el_antenna_idxs=[9, 8, 5, 4],
                              detector_type="range_doppler_ground_detector", detector_params=PRECISE_GROUND)
    for f in range(3):
        d[f"ground_pc_f{f}"] = pcg.process(seq[f])
    # non-finite samples: an end antenna (Hann weight exactly 0) and a middle one -- where is the 3-D chain's output finite?
    dbs = RangeAngleProcessorDBSEnhanced(cm)
    for tag, ant in (("inf_ant0", 0), ("inf_ant5", 5)):
        cube = synth.synth_cube(3).copy()
        cube[ant, 17, 9] = np.inf
        f3 = dbs.compute_3d_windowed_fft(cube)
        d[f"{tag}_finite_count"] = np.array(int(np.isfinite(f3).sum()))
        d[f"{tag}_nan_count"] = np.array(int(np.isnan(f3).sum()))
        d[f"{tag}_size"] = np.array(f3.size)
    np.savez_compressed(os.path.join(HERE, "detectors_rd.npz"), **d)
    print("detectors_rd.npz:", {k: (v.shape if v.ndim else v.item()) for k, v in d.items()})


# shipped cfg files with non-power-of-two planes (shape = virtual antennas x samples x loops), the antenna lists used with them
NP2_CASES = (("1843_RaGNNarok_UAV_10m.cfg", (8, 254, 50), [0, 1, 2, 3], [4, 5, 6, 7]),
             ("1843_RadVel_5Hz.cfg", (8, 63, 127), [0, 1, 2, 3, 4, 5, 6, 7], [1, 5]),
             ("RadSAR.cfg", (12, 100, 100), [0, 1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11]),
             ("1843_vel_nav.cfg", (4, 127, 32), [0, 1, 2, 3], []))
NP2_SEEDS = (411, 412)


def gen_detectors_np2():
    """Detector and point-cloud outputs of the reference on four more shipped cfg shapes, none a power of two (round 3 had one
    such fixture, 12 x 63 x 70): PointCloudGenerator with the CA-CFAR 2-D detector, the GUI's OS-CFAR 2-D and the YAML
    sequential detector, two seeded cubes per shape."""
    d = {}
    for cfg, shape, az, el in NP2_CASES:
        with open(os.path.join(REF, "configs", cfg)) as f:
            cm = load_cm(f.read())
        tag = "x".join(str(x) for x in shape)
        assert (cm.num_rx_antennas * (cm.num_tx_antennas if cm.virtual_antennas_enabled else 1), cm.get_num_adc_samples(0),
                cm.frameCfg_loops) == shape, (cfg, shape)
        for seed in NP2_SEEDS:
            cube = synth.synth_cube(seed, shape)
            pcg = PointCloudGenerator(cm, az_antenna_idxs=az, el_antenna_idxs=el, detector_type="range_doppler_detector_2d",
                                      detector_params={"cfar_type": "ca_cfar_2d",
                                                       "cfar_params": {"num_train": (4, 4), "num_guard": (2, 2), "pfa": 1e-5}})
            d[f"{tag}_s{seed}_pc"] = pcg.process(cube)
            d[f"{tag}_s{seed}_dets"] = np.asarray(pcg.detector.dets)
            d[f"{tag}_s{seed}_os2d"] = RangeDopplerDetector2D(cm, cfar_type="os_cfar_2d", cfar_params=YAML_OS2D).process(cube)
            d[f"{tag}_s{seed}_seq"] = RangeDopplerDetectorSequential(cm, **YAML_SEQ).process(cube)
    np.savez_compressed(os.path.join(HERE, "detectors_np2.npz"), **d)
    print("detectors_np2.npz:", {k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "detectors_np2":
        gen_detectors_np2()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "doppler_azimuth":
        gen_doppler_azimuth()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "detectors_rd":
        gen_detectors_rd()
        sys.exit(0)
    gen_cfgs()
    gen_small_chain()
    gen_frames_256()
    gen_cfar_known()
    gen_bartlett()
    gen_doppler_azimuth()
    gen_detectors_rd()
    gen_detectors_np2()
