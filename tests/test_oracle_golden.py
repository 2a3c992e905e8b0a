"""Pin oracle/oracle_np.py to the fixtures generated from the imported reference
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

from mmwave_radar_processing_amd import synth
from oracle import oracle_np as O

from conftest import GOLDEN

TIGHT = dict(rtol=1e-12, atol=0)


def close(a, b, scale_rtol=1e-12):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape
    m = np.max(np.abs(b)) if b.size else 1.0
    assert np.max(np.abs(a - b)) <= scale_rtol * max(m, 1e-300) if b.size else True


def test_cfg_scalars_all_shipped_configs():
    with open(os.path.join(GOLDEN, "cfg_scalars.json")) as f:
        table = json.load(f)
    assert len(table) == 26
    for name, ent in table.items():
        got = O.cfg_scalars("\n".join(ent["lines"]) + "\n")
        for k, v in ent["expect"].items():
            assert got[k] == v, (name, k, got[k], v)   # exact: same float64 expression


def test_small_pow2_chain(golden):
    g = golden("small_chain.npz")
    sc = O.cfg_scalars(synth.synth_cfg_text(num_samples=32, num_loops=16))
    cube = synth.synth_cube(101, (12, 32, 16))
    close(O.range_doppler(cube), g["p2_rd"])
    close(O.range_doppler_process(cube, rx_idx=3, return_magnitude=True), g["p2_rd_mag_rx3"])
    rb, vb = O.rd_bins(sc)
    np.testing.assert_array_equal(rb, g["p2_range_bins"])
    np.testing.assert_array_equal(vb, g["p2_vel_bins"])
    close(O.range_profile(cube, 5), g["p2_range_profile_c5"])
    close(O.range_angle(cube, 64, chirp_idx=2), g["p2_ra_all"])
    close(O.range_angle(cube, 64, chirp_idx=0, rx_antennas=[0, 3, 4, 7]), g["p2_ra_sub"])
    close(O.range_angle(cube, 64, chirp_idx=1, rx_antennas=[1, 2], perform_windowing=False), g["p2_ra_nowin"])
    ph, ab = O.angle_tables(64)
    np.testing.assert_array_equal(ph, g["p2_phase_shifts"])
    np.testing.assert_array_equal(ab, g["p2_angle_bins"])
    np.testing.assert_array_equal(O.ra_range_bins(sc), g["p2_ra_range_bins"])
    f3 = O.fft3d_windowed(cube, 64)
    close(f3, g["p2_fft3d"])
    dbs_bins = np.linspace(ab[0], ab[-1], 40)
    np.testing.assert_array_equal(dbs_bins, g["p2_dbs_angle_bins"])
    close(O.dbs_sharpen(np.abs(f3), g["p2_dbs_vel"], ab, dbs_bins, vb), g["p2_dbs"])


def test_sample_cfg_shape_non_pow2(golden):
    g = golden("small_chain.npz")
    with open(os.path.join(GOLDEN, "cfg_scalars.json")) as f:
        ent = json.load(f)["6843_RadVel_ods_20Hz.cfg"]
    sc = O.cfg_scalars("\n".join(ent["lines"]))
    assert (sc["num_rx"], sc["num_tx"], sc["num_samples"], sc["loops"]) == (4, 3, 63, 70)
    raw = synth.synth_raw_cube(202, 4, 3, 63, 70)
    virt = O.virtual_array_reformat(raw, sc["num_rx"], sc["frame_start"], sc["frame_end"], sc["loops"])
    assert virt.dtype == np.complex128 and virt.shape == (12, 63, 70)
    np.testing.assert_array_equal(virt[:, ::7, ::9], g["np2_virt_sample"])
    np.testing.assert_array_equal(virt, synth.synth_cube(202, (12, 63, 70)))
    rd = O.range_doppler(virt)
    close(rd[:, ::3, ::5], g["np2_rd_sample"])
    close(O.fft3d_windowed(virt, 64)[::4, ::3, ::5], g["np2_fft3d_sample"])
    _, mag, dets, _, _ = O.rd_detect_2d(virt)
    np.testing.assert_array_equal(dets, g["np2_dets"])
    close(mag, g["np2_mag0"])
    raw3 = synth.synth_raw_cube(303, 4, 3, 16, 8)
    np.testing.assert_array_equal(O.virtual_array_reformat(raw3, 4, 0, 2, 8), g["var_small"])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_headline_shape_frames(golden, seed):
    g = golden("frames_256.npz")
    sc = O.cfg_scalars(synth.SYNTH_CFG_256x128x12)
    cube = synth.synth_cube(seed)
    pc, dets, az_i, el_i = O.point_cloud(cube, sc, list(range(8)), [8, 9, 10, 11])
    np.testing.assert_array_equal(dets, g[f"s{seed}_dets"])
    assert dets.dtype == np.int64 and dets.shape[0] > 50
    np.testing.assert_allclose(pc, g[f"s{seed}_pc"], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(g["angle_bins"][az_i], g[f"s{seed}_az"])
    np.testing.assert_array_equal(g["angle_bins"][el_i], g[f"s{seed}_el"])
    raw, mag, _, thr, noise = O.rd_detect_2d(cube)
    close(raw[:, ::8, ::8], g[f"s{seed}_rd_sample"])
    np.testing.assert_array_equal(thr[::4, ::4], g[f"s{seed}_thr_sample"])
    np.testing.assert_array_equal(noise[::4, ::4], g[f"s{seed}_noise_sample"])
    if seed < 2:
        close(mag, g[f"s{seed}_mag0"])
    close(O.fft3d_windowed(cube)[::4, ::8, ::8], g[f"s{seed}_fft3d_sample"])


def test_noise_only_frame_has_no_detections(golden):
    g = golden("frames_256.npz")
    _, _, dets, _, _ = O.rd_detect_2d(synth.synth_cube(77, num_targets=0))
    np.testing.assert_array_equal(dets, g["noise77_dets"])
    assert dets.shape == (0, 2)


def test_cfar_known_answers(golden):
    """Replay of the reference's tests/verify_detectors_manual.py:15-89 facts."""
    g = golden("cfar_known.npz")
    np.random.seed(42)
    x = np.random.exponential(scale=1.0, size=100)
    x[50] = 10.0
    cases = dict(ca=O.ca_cfar_1d(x, 10, 2, 1e-3), go=O.go_cfar_1d(x, 10, 2, 1e-3),
                 so=O.so_cfar_1d(x, 10, 2, 1e-3), os=O.os_cfar_1d(x, 10, 2, 0.75, 5.0),
                 os_ascalled=O.os_cfar_1d(x, 10, 2, 15, 5.0))
    for name, (thr, noise, dets) in cases.items():
        np.testing.assert_array_equal(np.array(dets, dtype=np.int64), g[f"1d_{name}_dets"])
        np.testing.assert_array_equal(thr, g[f"1d_{name}_thr"])
        np.testing.assert_array_equal(noise, g[f"1d_{name}_noise"])
    assert cases["ca"][2] == [50] and cases["go"][2] == [50] and cases["so"][2] == [50]
    assert cases["os"][2] == [50] and cases["os_ascalled"][2] == []
    assert O.alpha_ca(20, 1e-3) == 8.250750892455088 == float(g["alpha_20_1e-3"])
    assert cases["ca"][0][50] == 6.811747051776046
    assert np.all(np.isfinite(cases["ca"][0][12:88])) and np.all(np.isinf(cases["ca"][0][:12]))
    X = np.random.exponential(scale=1.0, size=(50, 50))
    X[25, 25] = 15.0
    c2 = dict(ca=O.ca_cfar_2d(X, (5, 5), (2, 2), 1e-4), os=O.os_cfar_2d(X, (5, 5), (2, 2), 0.8, 5.0),
              os_yaml=O.os_cfar_2d(X, [5, 5], [3, 2], 0.7, 2))
    for name, (thr, noise, dets) in c2.items():
        np.testing.assert_array_equal(np.array(dets, dtype=np.int64).reshape(-1, 2), g[f"2d_{name}_dets"])
        np.testing.assert_array_equal(thr, g[f"2d_{name}_thr"])
        np.testing.assert_array_equal(noise, g[f"2d_{name}_noise"])
    assert c2["ca"][2] == [(25, 25)] and c2["os"][2] == [(8, 31), (25, 25)]
    assert O.alpha_ca(200, 1e-4) == 9.425709610179922
    assert c2["ca"][0][25, 25] == 9.66971770893241
    assert O.ca_cfar_2d(np.ones((5, 5)), (4, 4), (2, 2), 1e-5)[2] == []
    with pytest.raises(ValueError):
        O.ca_cfar_1d(np.ones((3, 3)), 1, 1, 1e-3)
    with pytest.raises(ValueError):
        O.ca_cfar_2d(np.ones(9), (1, 1), (1, 1), 1e-3)


def test_bartlett_contraction(golden):
    g = golden("bartlett_small.npz")
    rng = np.random.default_rng(404)
    frames, S, chirps = 2, 64, 24
    hist = rng.standard_normal((frames, S, chirps)) + 1j * rng.standard_normal((frames, S, chirps))
    geom = rng.uniform(-0.02, 0.02, (frames, 3, chirps))
    X = hist.transpose((1, 0, 2)).reshape(S, -1)
    P = geom.transpose((1, 0, 2)).reshape(3, -1)
    d = O.steering_dirs(g["az"], g["el"])
    np.testing.assert_array_equal(d, g["d"])
    close(O.bartlett_response(X, P, d, float(g["lambda_m"])), g["out"])


def test_capon_single_source_peak():
    """No upstream oracle (parity unpinned): analytic single-source sanity of the build's own definition."""
    rng = np.random.default_rng(5)
    V, R, K = 12, 4, 128
    th0 = 0.3
    a = np.exp(-1j * np.pi * np.arange(V) * np.sin(th0))
    s = rng.standard_normal((R, K)) + 1j * rng.standard_normal((R, K))
    X = a[:, None, None] * s[None] * 10 + 0.1 * (rng.standard_normal((V, R, K)) + 1j * rng.standard_normal((V, R, K)))
    th = np.linspace(-1.2, 1.2, 241)
    P = O.capon_spectrum(X, th)
    assert np.all(np.abs(th[np.argmax(P, axis=1)] - th0) <= 0.011)


def test_doppler_azimuth_coarse_path(golden):
    g = golden("doppler_azimuth.npz")
    sc = O.cfg_scalars(synth.synth_cfg_text(num_samples=32, num_loops=16))
    cube = synth.synth_cube(101, (12, 32, 16))
    close(O.doppler_azimuth(cube, sc), g["std_all"])
    close(O.doppler_azimuth(cube, sc, rx_antennas=[4, 5, 8, 9], range_window=[0.9, 2.0], shift_angle=False),
          g["std_sub_win"])
    with open(os.path.join(GOLDEN, "cfg_scalars.json")) as f:
        sc2 = O.cfg_scalars("\n".join(json.load(f)["6843_RadVel_ods_20Hz.cfg"]["lines"]))
    virt = synth.synth_cube(202, (12, 63, 70))
    close(O.doppler_azimuth(virt, sc2, rx_antennas=[4, 5, 8, 9], range_window=[0.9, 2.0], shift_angle=False,
                            valid_angle_range=(-1.04719755, 1.04719755), standard_geometry=False), g["ods_sub"])


def test_doppler_azimuth_precise_path(golden):
    """use_precise_fft=True: two scipy ZoomFFT calls inside the reference (doppler_azimuth_resp.py:130-294)."""
    g = golden("doppler_azimuth.npz")
    sc = O.cfg_scalars(synth.synth_cfg_text(num_samples=32, num_loops=16))
    cube = synth.synth_cube(101, (12, 32, 16))
    for tag, vr, kw in (("default", [-0.25, 0.25], {}), ("pos_only", [0.3, 1.2], {}), ("narrow", [-0.05, 0.02], {}),
                        ("clamped", [-50.0, 50.0], {"shift_angle": False}),
                        ("neg_sub", [-1.0, -0.2], {"rx_antennas": [4, 5, 8, 9], "range_window": [0.9, 2.0]})):
        resp, bins = O.doppler_azimuth_precise(cube, sc, vel_range=vr, **kw)
        np.testing.assert_allclose(bins, g["precise_" + tag + "_bins"], rtol=0, atol=1e-15)
        close(resp, g["precise_" + tag], 1e-10)
    assert np.all(g["precise_narrow"][:16] == 0) and np.all(g["precise_narrow"][16:] > 0)   # zero rows of the short half
    with open(os.path.join(GOLDEN, "cfg_scalars.json")) as f:
        sc2 = O.cfg_scalars("\n".join(json.load(f)["6843_RadVel_ods_20Hz.cfg"]["lines"]))
    virt = synth.synth_cube(202, (12, 63, 70))
    resp, bins = O.doppler_azimuth_precise(virt, sc2, rx_antennas=[4, 5, 8, 9], range_window=[0.9, 2.0],
                                           valid_angle_range=(-1.04719755, 1.04719755), standard_geometry=False)
    np.testing.assert_allclose(bins, g["precise_ods_bins"], rtol=0, atol=1e-15)
    close(resp, g["precise_ods"], 1e-10)


def test_range_zoom_matches_reference_zoomfft(golden):
    g = golden("doppler_azimuth.npz")
    sc = O.cfg_scalars(synth.synth_cfg_text(num_samples=32, num_loops=16))
    cube = synth.synth_cube(101, (12, 32, 16))
    z, zb = O.range_zoom(cube, sc, 0.6, 1.9, chirp_idx=3)
    close(z, g["zoom_mag"], 1e-11)
    np.testing.assert_array_equal(zb, g["zoom_bins"])


# parameters of tests/golden/make_golden.py::gen_detectors_rd (gui_configs/processor_params.yaml:40-86 and variants)
YAML_OS2D = dict(num_train=(5, 5), num_guard=(3, 2), rho=0.7, alpha=2)
YAML_SEQ = ("os_cfar_1d", {"num_train": 5, "num_guard": 3, "rho": 0.6, "alpha": 2},
            "os_cfar_1d", {"num_train": 5, "num_guard": 2, "rho": 0.7, "alpha": 3})
GOSO_SEQ = ("go_cfar_1d", {"num_train": 8, "num_guard": 2, "pfa": 1e-3},
            "so_cfar_1d", {"num_train": 6, "num_guard": 2, "pfa": 1e-4})
GROUND = {
    "yaml": ("os_cfar_1d", {"num_train": 12, "num_guard": 4, "rho": 0.5, "alpha": 15},
             {"min_altitude_m": 0.25, "zoom_search_region_m": 0.2, "altitude_search_limit_m": 0.4, "range_bias": 0.0,
              "precise_est_enabled": False}),
    "coarse": ("os_cfar_1d", {"num_train": 12, "num_guard": 4, "rho": 0.5, "alpha": 6},
               {"min_altitude_m": 0.6, "zoom_search_region_m": 0.2, "altitude_search_limit_m": 0.6, "range_bias": 0.0,
                "precise_est_enabled": False}),
    "precise": ("os_cfar_1d", {"num_train": 16, "num_guard": 4, "rho": 0.5, "alpha": 12},
                {"min_altitude_m": 0.6, "zoom_search_region_m": 0.2, "altitude_search_limit_m": 0.6, "range_bias": 0.03,
                 "precise_est_enabled": True}),
}


def _rd_cases():
    sc256 = O.cfg_scalars(synth.SYNTH_CFG_256x128x12)
    with open(os.path.join(GOLDEN, "cfg_scalars.json")) as fh:
        ods = json.load(fh)["6843_RadVel_ods_20Hz.cfg"]
    sc_ods = O.cfg_scalars("\n".join(ods["lines"]))
    return [(f"s{s}", sc256, synth.synth_cube(s)) for s in (0, 1, 2, 3)] + [("np2", sc_ods, synth.synth_cube(202, (12, 63, 70)))]


def test_sequential_and_os2d_detectors_on_range_doppler_data():
    """Reference-generated detections of RangeDopplerDetectorSequential (YAML OS parameters and a GO / SO pairing) and of
    RangeDopplerDetector2D with the YAML's os_cfar_2d on four headline frames and the (12, 63, 70) cube: oracle == reference,
    values and order."""
    g = np.load(os.path.join(GOLDEN, "detectors_rd.npz"))
    total = 0
    for tag, _, cube in _rd_cases():
        np.testing.assert_array_equal(O.rd_detect_sequential(cube, *YAML_SEQ), g[f"{tag}_seq_yaml"])
        np.testing.assert_array_equal(O.rd_detect_sequential(cube, *GOSO_SEQ), g[f"{tag}_seq_goso"])
        np.testing.assert_array_equal(O.rd_detect_2d_os(cube, **YAML_OS2D), g[f"{tag}_os2d_yaml"])
        total += g[f"{tag}_seq_yaml"].shape[0] + g[f"{tag}_os2d_yaml"].shape[0]
    assert total > 2000


# tests/golden/make_golden.py::gen_detectors_np2: shipped cfg files with non-power-of-two planes
NP2_CASES = (("1843_RaGNNarok_UAV_10m.cfg", (8, 254, 50), [0, 1, 2, 3], [4, 5, 6, 7]),
             ("1843_RadVel_5Hz.cfg", (8, 63, 127), [0, 1, 2, 3, 4, 5, 6, 7], [1, 5]),
             ("RadSAR.cfg", (12, 100, 100), [0, 1, 2, 3, 4, 5, 6, 7], [8, 9, 10, 11]),
             ("1843_vel_nav.cfg", (4, 127, 32), [0, 1, 2, 3], []))
NP2_SEEDS = (411, 412)


def np2_cfg_text(cfg):
    with open(os.path.join(GOLDEN, "cfg_scalars.json")) as fh:
        return "\n".join(json.load(fh)[cfg]["lines"])


def test_detectors_on_more_non_power_of_two_shapes():
    """Four more shipped cfg shapes (254 x 50, 63 x 127, 100 x 100, 127 x 32; round 3 pinned 63 x 70 only), two cubes each, run by
    the imported reference: CA-CFAR detections + point cloud, the GUI's OS-CFAR 2-D and the YAML sequential detector --
    oracle == reference."""
    g = np.load(os.path.join(GOLDEN, "detectors_np2.npz"))
    total = 0
    for cfg, shape, az, el in NP2_CASES:
        sc = O.cfg_scalars(np2_cfg_text(cfg))
        tag = "x".join(str(x) for x in shape)
        for seed in NP2_SEEDS:
            cube = synth.synth_cube(seed, shape)
            pc, dets, _, _ = O.point_cloud(cube, sc, az, el)
            np.testing.assert_array_equal(dets, g[f"{tag}_s{seed}_dets"])
            close(pc, g[f"{tag}_s{seed}_pc"], 1e-12)
            np.testing.assert_array_equal(O.rd_detect_2d_os(cube, **YAML_OS2D), g[f"{tag}_s{seed}_os2d"])
            np.testing.assert_array_equal(O.rd_detect_sequential(cube, *YAML_SEQ), g[f"{tag}_s{seed}_seq"])
            total += dets.shape[0]
    assert total > 300


def test_ground_detector_sequence_with_altimeter_state():
    """RangeDopplerGroundDetector over a 5-frame sequence (the altimeter keeps its last altitude), reset(), one more frame:
    detections and the altitude track of the oracle == reference, for the YAML parameters (never locks on: the ground return
    lies beyond its search limit), a coarse lock and a zoom-FFT (precise) lock."""
    g = np.load(os.path.join(GOLDEN, "detectors_rd.npz"))
    sc = O.cfg_scalars(synth.SYNTH_CFG_256x128x12)
    seq = synth.synth_ground_sequence(606, 5)
    for name, (vel_kind, vel_params, alt_params) in GROUND.items():
        alt = O.Altimeter(sc, **alt_params)
        track = []
        for f in range(5):
            dets, a = O.rd_detect_ground(seq[f], alt, sc, vel_kind, vel_params, alt_params)
            np.testing.assert_array_equal(dets, g[f"ground_{name}_f{f}"])
            track.append(a)
        np.testing.assert_allclose(track, g[f"ground_{name}_alt"], rtol=0, atol=1e-12)
        alt.reset()
        dets, a = O.rd_detect_ground(seq[3], alt, sc, vel_kind, vel_params, alt_params)
        np.testing.assert_array_equal(dets, g[f"ground_{name}_after_reset_f3"])
        np.testing.assert_allclose(a, g[f"ground_{name}_after_reset_alt"], rtol=0, atol=1e-12)
    assert g["ground_precise_f0"].shape[0] > 0 and g["ground_precise_alt"][4] > g["ground_precise_alt"][0]


def test_non_finite_sample_poisons_the_reference_chain():
    """A single inf sample -- in an end antenna (Hann weight exactly 0: 0 * inf = NaN) or a middle one -- leaves no finite
    value in the reference's 3-D cube; the oracle restates that."""
    g = np.load(os.path.join(GOLDEN, "detectors_rd.npz"))
    for tag, ant in (("inf_ant0", 0), ("inf_ant5", 5)):
        cube = synth.synth_cube(3).copy()
        cube[ant, 17, 9] = np.inf
        with np.errstate(all="ignore"):
            f3 = O.fft3d_windowed(cube, 64)
        assert int(np.isfinite(f3).sum()) == int(g[f"{tag}_finite_count"]) == 0
        assert f3.size == int(g[f"{tag}_size"])
