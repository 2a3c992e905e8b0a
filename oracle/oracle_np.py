"""CPU oracle: float64 NumPy restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mmwave_radar_processing_amd/`` may
import this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` do, and only as the checker / the timed
CPU baseline -- never as the thing shipped.

Parity status: PINNED.  Every function below is asserted equal to the imported
reference (davidmhunt/mmwave_radar_processing @ /root/reference, run in the
build container) through the fixtures in ``tests/golden/`` written by
``tests/golden/make_golden.py``; ``tests/test_oracle_golden.py`` re-checks this
module against those fixtures on every run.  Exception: ``capon_*`` has no
upstream counterpart -> "parity unpinned" for those two.

Each function cites the reference file:line it restates (paths relative to
``mmwave_radar_processing/`` in the reference).  All arithmetic is
float64/complex128 like the reference.
"""
from __future__ import annotations

import numpy as np
from numpy.lib.stride_tricks import sliding_window_view

C_LIGHT = 299792458.0  # scipy.constants.c, used at config_managers/cfgManager.py:119


# --------------------------------------------------------------------------
# a2  ConfigManager scalars -- config_managers/cfgManager.py:210-239,113-169,266-360
# --------------------------------------------------------------------------
def cfg_scalars(cfg_text: str) -> dict:
    """Derived scalars of a TI mmWave cfg (lines containing '%' are skipped, :234)."""
    prof = None
    frame = None
    rx_mask = tx_mask = None
    n_chirp_cfgs = 0
    for line in cfg_text.splitlines():
        if "%" in line:
            continue
        tok = line.strip("\n").split(" ")
        if tok[0] == "channelCfg":          # :266-279
            rx_mask, tx_mask = int(tok[1]), int(tok[2])
        elif tok[0] == "profileCfg" and prof is None:   # :307-333 (first profile wins slot 0)
            prof = dict(start_ghz=float(tok[2]), idle_us=float(tok[3]), ramp_end_us=float(tok[5]),
                        slope=float(tok[8]), samples=int(tok[10]), ksps=int(tok[11]))
        elif tok[0] == "chirpCfg":
            n_chirp_cfgs += 1
        elif tok[0] == "frameCfg":          # :352-360
            frame = dict(start=int(tok[1]), end=int(tok[2]), loops=int(tok[3]),
                         frames=int(tok[4]), period_ms=float(tok[5]))
    S = prof["samples"]
    fs = prof["ksps"] * 1e3
    slope = prof["slope"] * (1e6 / 1e-6)
    range_res = (C_LIGHT * fs) / (2 * slope * S)                   # :119-122
    range_max = (C_LIGHT * fs) / (2 * slope)                       # :129-130
    nfft = np.power(2, np.ceil(np.log2(S)))
    range_bin = (C_LIGHT * fs) / (2 * slope * nfft)                # :124-127
    lam = C_LIGHT / (prof["start_ghz"] * 1e9)                      # :139
    cpl = frame["end"] - frame["start"] + 1
    tc_us = prof["ramp_end_us"] + prof["idle_us"]
    vel_res = lam / (2 * tc_us * cpl * 1e-6 * float(frame["loops"]))   # :151-153
    vel_max = lam / (4 * tc_us * cpl * 1e-6)                           # :156
    return dict(
        num_rx=bin(rx_mask).count("1"), num_tx=bin(tx_mask).count("1"),
        num_samples=S, loops=frame["loops"], chirps_per_loop=cpl,
        frame_start=frame["start"], frame_end=frame["end"],
        range_res_m=range_res, range_max_m=range_max, range_bin_size_m=float(range_bin),
        vel_res_m_s=vel_res, vel_max_m_s=vel_max,
        virtual_antennas_enabled=cpl > 1,                          # :162-167
    )


def rd_bins(sc: dict):
    """range_bins / vel_bins of RangeDopplerProcessor.configure -- processors/range_doppler_resp.py:33-47."""
    vel = np.arange(-sc["vel_max_m_s"], sc["vel_max_m_s"] - sc["vel_res_m_s"] + 1e-3, sc["vel_res_m_s"])
    rng = np.arange(0, sc["range_max_m"] - sc["range_res_m"] / 2 + 1e-3, sc["range_res_m"])
    return rng, vel


def angle_tables(num_angle_bins: int):
    """phase_shifts / angle_bins -- processors/range_angle_resp.py:38-48 (same at point_cloud_generator.py:95-105)."""
    A = num_angle_bins
    ph = np.arange(np.pi, -np.pi - 2 * np.pi / (A - 1), -2 * np.pi / (A - 1))
    ph[-1] = -np.pi
    return ph, np.arcsin(ph / np.pi)


def ra_range_bins(sc: dict):
    """RangeAngleProcessor.range_bins (+1 mm offset) -- processors/range_angle_resp.py:31-34."""
    return np.arange(0, sc["range_max_m"] - sc["range_res_m"] / 2, sc["range_res_m"]) + 1e-3


# --------------------------------------------------------------------------
# a3  VirtualArrayReformatter.process -- processors/virtual_array_reformater.py:44-65
# --------------------------------------------------------------------------
def virtual_array_reformat(raw: np.ndarray, num_rx: int, frame_start: int, frame_end: int,
                           loops: int) -> np.ndarray:
    cfg_ids = np.arange(frame_start, frame_end + 1)
    per_chirp = np.tile(cfg_ids, loops)
    out = np.zeros((num_rx * len(cfg_ids), raw.shape[1], loops), dtype=complex)
    for i, cid in enumerate(cfg_ids):
        out[i * num_rx:(i + 1) * num_rx] = raw[0:num_rx][:, :, per_chirp == cid]
    return out


# --------------------------------------------------------------------------
# a4  RangeProcessor.coarse_fft -- processors/range_resp.py:32-57
# --------------------------------------------------------------------------
def range_profile(cube: np.ndarray, chirp_idx: int = 0) -> np.ndarray:
    x = cube[:, :, chirp_idx] * np.hanning(cube.shape[1])
    return np.mean(np.abs(np.fft.fft(x, axis=1)), axis=0)


def range_zoom(cube, sc, range_start_m, range_stop_m, chirp_idx=0):
    """RangeProcessor.zoom_fft -- processors/range_resp.py:59-102.  scipy's ZoomFFT(n, [f1, f2], fs) is the DFT on
    the arc f_k = f1 + k (f2 - f1)/n, k < n (endpoint=False); restated as that direct sum (float64)."""
    x = cube[:, :, chirp_idx] * np.hanning(cube.shape[1])
    S = x.shape[1]
    fs = 1 / sc["range_res_m"]
    f1 = range_start_m * fs / sc["range_max_m"]
    f2 = range_stop_m * fs / sc["range_max_m"]
    fk = f1 + np.arange(S) * (f2 - f1) / S
    kern = np.exp(-2j * np.pi * np.outer(fk / fs, np.arange(S)))        # [m, n]
    return np.mean(np.abs(x @ kern.T), axis=0), np.linspace(range_start_m, range_stop_m, S)


# --------------------------------------------------------------------------
# a5/a6  RangeDopplerProcessor -- processors/range_doppler_resp.py:49-110
# --------------------------------------------------------------------------
def range_doppler(cube: np.ndarray) -> np.ndarray:
    """Complex128 ``[V,S,C]``: Hann x Hann, fft2 over (sample, chirp), fftshift Doppler only."""
    xw = cube * np.hanning(cube.shape[1])[None, :, None]
    xw = xw * np.hanning(cube.shape[2])[None, None, :]
    return np.fft.fftshift(np.fft.fft2(xw, axes=(-2, -1)), axes=-1)


def range_doppler_process(cube, rx_idx=0, return_magnitude=True):
    """RangeDopplerProcessor.process incl. the slice-after-compute quirk (:108-110)."""
    r = range_doppler(cube)
    if return_magnitude:
        r = np.abs(r)
    if rx_idx >= 0:
        r = r[rx_idx]
    return r


# --------------------------------------------------------------------------
# a8  RangeAngleProcessor.process -- processors/range_angle_resp.py:55-122
# --------------------------------------------------------------------------
def range_angle(cube, num_angle_bins=64, chirp_idx=0, rx_antennas=(), perform_windowing=True):
    rx = np.asarray(rx_antennas)
    x = cube
    if perform_windowing:   # windows over ALL rx before the subset is taken (:96-101)
        x = x * np.hanning(x.shape[1])[None, :, None]
        x = x * np.hanning(x.shape[0])[:, None, None]
    if rx.size > 0:
        x = x[rx]
    data = np.zeros((cube.shape[1], num_angle_bins), dtype=complex)
    data[:, :x.shape[0]] = x[:, :, chirp_idx].T
    return np.abs(np.fft.fftshift(np.fft.fft2(data, axes=(0, 1)), axes=1))


# --------------------------------------------------------------------------
# a9/a10  RangeAngleProcessorDBSEnhanced -- processors/range_angle_resp_dbs_enhanced.py:137-263
# --------------------------------------------------------------------------
def fft3d_windowed(cube: np.ndarray, num_angle_bins: int = 64) -> np.ndarray:
    """Complex128 ``[A,S,C]``: range FFT -> Doppler FFT+shift -> angle FFT (zero-pad V->A)+shift."""
    V, S, C = cube.shape
    r = np.fft.fft(cube * np.hanning(S)[None, :, None], axis=-2)
    rd = np.fft.fftshift(np.fft.fft(r * np.hanning(C)[None, None, :], axis=-1), axes=-1)
    pad = np.zeros((num_angle_bins, S, C), dtype=complex)
    pad[:V] = rd * np.hanning(V)[:, None, None]
    return np.fft.fftshift(np.fft.fft(pad, axis=0), axes=0)


def dbs_sharpen(mag_asc, velocity_ned, angle_bins_no_dbs, angle_bins_dbs, vel_bins):
    """perform_dbs_sharpen (:216-263): per output angle pick [nearest angle, :, nearest vel]."""
    out = np.zeros((len(angle_bins_dbs), mag_asc.shape[1]))
    for i, ang in enumerate(angle_bins_dbs):
        r = np.array([np.cos(ang), np.sin(ang), 0.0])
        dop = -1 * np.dot(r / np.linalg.norm(r), velocity_ned)         # get_dop_vel :200-214
        vb = np.argmin(np.abs(vel_bins - dop))
        ab = np.argmin(np.abs(angle_bins_no_dbs - ang))
        out[i] = mag_asc[ab, :, vb]
    return out.T


# --------------------------------------------------------------------------
# f-2  DopplerAzimuthProcessor.process, coarse path -- processors/doppler_azimuth_resp.py:84-128,296-334,419-491
# --------------------------------------------------------------------------
def doppler_azimuth(cube, sc, num_angle_bins=64, rx_antennas=(), range_window=(), shift_angle=True,
                    valid_angle_range=(np.deg2rad(-60), np.deg2rad(60)), standard_geometry=True):
    rx = np.asarray(rx_antennas)
    x = cube[rx] if rx.size > 0 else cube                                   # :462-466
    V, S, C = x.shape
    xw = x * np.hanning(S)[None, :, None]
    xw = xw * np.hanning(C)[None, None, :]
    if standard_geometry and sc["virtual_antennas_enabled"]:                # :97-100
        xw = xw * np.hanning(V)[:, None, None]
    rng_bins, _ = rd_bins(sc)
    rw = np.asarray(range_window, dtype=float)
    if rw.size == 0:
        rw = np.array([0, sc["range_max_m"]])
    r = np.fft.fft(xw, axis=1)[:, (rng_bins >= rw[0]) & (rng_bins <= rw[1]), :]   # :119-126
    data = np.zeros((r.shape[1], C, num_angle_bins), dtype=complex)
    data[:, :, :V] = np.transpose(r, (1, 2, 0))
    resp = np.abs(np.fft.fftshift(np.fft.fft2(data, axes=(1, 2)), axes=(1, 2) if shift_angle else (1)))   # :320-332
    _, abins = angle_tables(num_angle_bins)
    valid = (abins >= valid_angle_range[0]) & (abins <= valid_angle_range[1])
    return np.mean(resp[:, :, valid], axis=0)                               # :486-489


# f-2b  DopplerAzimuthProcessor.process(use_precise_fft=True) -- processors/doppler_azimuth_resp.py:130-294
def zoomed_vel_plan(sc, vel_range, min_zoom_fft_vel_span=0.1):
    """-> (zoomed_vel_bins, freq): the Doppler bins of the precise mode and, for each of them, the frequency in
    cycles per chirp that the reference's two ZoomFFT calls evaluate (NaN where it emits a zero row instead).

    The input range is NOT modified (the reference clamps/widens the caller's array in place, :234-246)."""
    vmax, vres = sc["vel_max_m_s"], sc["vel_res_m_s"]
    _, vel_bins = rd_bins(sc)
    nvb = vel_bins.size
    vr = np.array(vel_range, dtype=float)
    vr[0] = max(vr[0], -vmax)                                               # :234-235
    vr[1] = min(vr[1], vmax)
    spread = 2 * min_zoom_fft_vel_span                                      # :238-246
    if (vr[1] - vr[0]) < spread:
        d_hi, d_lo = abs(vr[1] - vmax), abs(vr[0] + vmax)
        if d_hi > d_lo:
            vr[1] = vr[0] + spread
        elif d_lo > d_hi:
            vr[0] = vr[1] - spread
    neg = np.linspace(vr[0], min(-1e-4, vr[1]), nvb if vr[0] <= 0 else 0, endpoint=False)   # :178-183
    pos = np.linspace(max(1e-4, vr[0]), vr[1], nvb if vr[1] > 0 else 0, endpoint=False)     # :186-191
    bins = np.concatenate((neg, pos))                                       # :194-201 (either part may be empty)
    fs = 1.0 / vres                                                         # :148
    freq = np.full(bins.size, np.nan)

    def seg(vals, offset, shift):
        m = vals.size
        if m > 0 and abs(vals.max() - vals.min()) > min_zoom_fft_vel_span:  # :254-255, :272-273
            lo, hi = vals.min() + shift, vals.max() + shift                 # :256-259 (negative half aliased up by 2 vmax)
            f1, f2 = lo * fs / vmax, hi * fs / vmax                         # :151-152
            fz = fs * 2                                                     # :155
            # scipy.signal.ZoomFFT(n=m, fn=[f1, f2], fs=fz): a = exp(2j pi f1/fz), w = exp(-2j pi (f2-f1)/(m fz)),
            # X[k] = sum_i x[i] a^-i w^(i k)
            freq[offset:offset + m] = f1 / fz + np.arange(m) * ((f2 - f1) / (m * fz))

    seg(bins[bins <= 0], 0, 2 * vmax)
    seg(bins[bins > 0], int(np.sum(bins <= 0)), 0.0)
    return bins, freq


def doppler_azimuth_precise(cube, sc, num_angle_bins=64, rx_antennas=(), range_window=(), shift_angle=True,
                            vel_range=(-0.25, 0.25), valid_angle_range=(np.deg2rad(-60), np.deg2rad(60)),
                            standard_geometry=True, min_zoom_fft_vel_span=0.1):
    """-> (resp [zoomed vel bins, valid angle bins], zoomed_vel_bins).  The zoom transform is evaluated directly,
    X[k] = sum_i x[i] exp(-2j pi i f_k), which is what ZoomFFT computes by Bluestein's algorithm."""
    rx = np.asarray(rx_antennas)
    x = cube[rx] if rx.size > 0 else cube
    V, S, C = x.shape
    xw = x * np.hanning(S)[None, :, None] * np.hanning(C)[None, None, :]
    if standard_geometry and sc["virtual_antennas_enabled"]:
        xw = xw * np.hanning(V)[:, None, None]
    rng_bins, _ = rd_bins(sc)
    rw = np.asarray(range_window, dtype=float)
    if rw.size == 0:
        rw = np.array([0, sc["range_max_m"]])
    r = np.fft.fft(xw, axis=1)[:, (rng_bins >= rw[0]) & (rng_bins <= rw[1]), :]
    bins, freq = zoomed_vel_plan(sc, vel_range, min_zoom_fft_vel_span)
    n_used = int(max(np.sum(bins <= 0), np.sum(bins > 0)))                  # data[:, :num_samples, :]  (:159-160)
    if n_used > C:
        raise ValueError("ZoomFFT defined for more chirps than the cube has")
    data = np.zeros((r.shape[1], n_used, num_angle_bins), dtype=complex)    # [range, chirp, padded antenna]  (:228-229)
    data[:, :, :V] = np.transpose(r, (1, 2, 0))[:, :n_used, :]
    f = np.where(np.isnan(freq), 0.0, freq)
    Z = np.exp(-2j * np.pi * np.outer(f, np.arange(n_used)))                # [bins, chirp]
    Z[np.isnan(freq)] = 0.0                                                 # zero rows (:267, :285)
    z = np.einsum("kc,sca->ska", Z, data)
    resp = np.abs(np.fft.fft(z, axis=2))                                    # :161-162
    if shift_angle:
        resp = np.fft.fftshift(resp, axes=2)                                # :290-291
    _, abins = angle_tables(num_angle_bins)
    valid = (abins >= valid_angle_range[0]) & (abins <= valid_angle_range[1])
    return np.mean(resp[:, :, valid], axis=0), bins


# --------------------------------------------------------------------------
# a13/a14  CFAR family -- detectors/base.py, ca_cfar.py, os_cfar.py, go_so_cfar.py
# --------------------------------------------------------------------------
def alpha_ca(n_train_cells, pfa):
    """detectors/base.py:154-169,281-293."""
    return n_train_cells * (pfa ** (-1.0 / n_train_cells) - 1.0)


def _finish_1d(x, est, thr_valid, half):
    L = len(x)
    thr = np.full(L, np.inf)
    noise = np.zeros(L)
    thr[half:half + len(est)] = thr_valid
    noise[half:half + len(est)] = est
    det = x > thr                                   # strict, base.py:61
    return thr, noise, np.where(det)[0].tolist()


def _empty_1d(x):
    return np.full(len(x), np.inf), np.zeros(len(x)), []


def ca_cfar_1d(x, num_train, num_guard, pfa):
    """CaCFAR1D -- detectors/ca_cfar.py:11-77 (N = 2*num_train)."""
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("Input x must be a 1D array.")
    w = 2 * (num_train + num_guard) + 1
    if len(x) < w:
        return _empty_1d(x)
    win = sliding_window_view(x, w)
    mask = np.ones(w, dtype=bool)
    mask[num_train:num_train + 2 * num_guard + 1] = False
    n = int(mask.sum())
    est = np.mean(win[:, mask], axis=1)
    return _finish_1d(x, est, alpha_ca(n, pfa) * est, num_train + num_guard)


def _sides_1d(x, num_train, num_guard):
    w = 2 * (num_train + num_guard) + 1
    win = sliding_window_view(x, w)
    return win[:, :num_train], win[:, num_train + 2 * num_guard + 1:]


def go_cfar_1d(x, num_train, num_guard, pfa):
    """GoCFAR1D -- detectors/go_so_cfar.py:11-70 (alpha from one side's N)."""
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("Input x must be a 1D array.")
    if len(x) < 2 * (num_train + num_guard) + 1:
        return _empty_1d(x)
    l, r = _sides_1d(x, num_train, num_guard)
    est = np.maximum(np.mean(l, axis=1), np.mean(r, axis=1))
    return _finish_1d(x, est, alpha_ca(num_train, pfa) * est, num_train + num_guard)


def so_cfar_1d(x, num_train, num_guard, pfa):
    """SoCFAR1D -- detectors/go_so_cfar.py:73-123."""
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("Input x must be a 1D array.")
    if len(x) < 2 * (num_train + num_guard) + 1:
        return _empty_1d(x)
    l, r = _sides_1d(x, num_train, num_guard)
    est = np.minimum(np.mean(l, axis=1), np.mean(r, axis=1))
    return _finish_1d(x, est, alpha_ca(num_train, pfa) * est, num_train + num_guard)


def os_k_rank(rho, n_train_cells):
    """k = clamp(int(rho*N), 1, N) -- detectors/os_cfar.py:25-27,131-132."""
    return max(1, min(int(rho * n_train_cells), n_train_cells))


def os_cfar_1d(x, num_train, num_guard, rho, alpha):
    """OsCFAR1D -- detectors/os_cfar.py:29-86."""
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("Input x must be a 1D array.")
    if len(x) < 2 * (num_train + num_guard) + 1:
        return _empty_1d(x)
    l, r = _sides_1d(x, num_train, num_guard)
    cells = np.concatenate((l, r), axis=1)
    k = os_k_rank(rho, 2 * num_train)
    est = np.partition(cells, k - 1, axis=1)[:, k - 1]
    return _finish_1d(x, est, alpha * est, num_train + num_guard)


def _mask_2d(num_train, num_guard):
    tr, td = num_train
    gr, gd = num_guard
    wr, wd = 2 * (tr + gr) + 1, 2 * (td + gd) + 1
    mask = np.ones((wr, wd), dtype=bool)
    mask[tr:tr + 2 * gr + 1, td:td + 2 * gd + 1] = False
    return mask, wr, wd


def _finish_2d(X, est, thr_valid, num_train, num_guard):
    R, D = X.shape
    thr = np.full((R, D), np.inf)
    noise = np.zeros((R, D))
    r0, d0 = num_train[0] + num_guard[0], num_train[1] + num_guard[1]
    thr[r0:r0 + est.shape[0], d0:d0 + est.shape[1]] = thr_valid
    noise[r0:r0 + est.shape[0], d0:d0 + est.shape[1]] = est
    rows, cols = np.where(X > thr)                  # row-major order, base.py:226-230
    return thr, noise, list(zip(rows.tolist(), cols.tolist()))


def ca_cfar_2d(X, num_train, num_guard, pfa):
    """CaCFAR2D -- detectors/ca_cfar.py:85-155 + BaseCFAR2D.detect base.py:208-230."""
    X = np.asarray(X)
    if X.ndim != 2:
        raise ValueError("Input X must be a 2D array.")
    mask, wr, wd = _mask_2d(num_train, num_guard)
    if X.shape[0] < wr or X.shape[1] < wd:
        return np.full(X.shape, np.inf), np.zeros(X.shape), []
    win = sliding_window_view(X, (wr, wd))
    n = int(mask.sum())
    est = np.sum(win * mask, axis=(2, 3)) / n
    return _finish_2d(X, est, alpha_ca(n, pfa) * est, num_train, num_guard)


def os_cfar_2d(X, num_train, num_guard, rho, alpha):
    """OsCFAR2D -- detectors/os_cfar.py:134-195."""
    X = np.asarray(X)
    if X.ndim != 2:
        raise ValueError("Input X must be a 2D array.")
    mask, wr, wd = _mask_2d(num_train, num_guard)
    if X.shape[0] < wr or X.shape[1] < wd:
        return np.full(X.shape, np.inf), np.zeros(X.shape), []
    cells = sliding_window_view(X, (wr, wd))[..., mask]
    k = os_k_rank(rho, int(mask.sum()))
    est = np.partition(cells, k - 1, axis=-1)[..., k - 1]
    return _finish_2d(X, est, alpha * est, num_train, num_guard)


# --------------------------------------------------------------------------
# a11/a12  RangeDopplerDetector2D -- processors/range_doppler_detection/*.py
# --------------------------------------------------------------------------
def rd_detect_2d(cube, num_train=(4, 4), num_guard=(2, 2), pfa=1e-5):
    """Returns (raw c128 [V,S,C], mag f64 [S,C] of rx 0 ONLY, dets int64 (N,2), thresholds, noise)."""
    raw = range_doppler(cube)                       # range_doppler_detector.py:72-76
    mag = np.abs(raw[0])                            # :78
    thr, noise, dets = ca_cfar_2d(mag, num_train, num_guard, pfa)
    d = np.array(dets, dtype=int) if dets else np.empty((0, 2), dtype=int)   # ..._2d.py:61-65
    return raw, mag, d, thr, noise


def rd_detect_2d_os(cube, num_train=(5, 5), num_guard=(3, 2), rho=0.7, alpha=2.0):
    """RangeDopplerDetector2D(cfar_type="os_cfar_2d") -- range_doppler_detector_2d.py:49-65 with detectors/os_cfar.py:97-195
    (the parameters of gui_configs/processor_params.yaml:40-47) -> dets int64 (N, 2)."""
    mag = np.abs(range_doppler(cube)[0])
    dets = os_cfar_2d(mag, num_train, num_guard, rho, alpha)[2]
    return np.array(dets, dtype=int) if dets else np.empty((0, 2), dtype=int)


_CFAR_1D = {"ca_cfar_1d": lambda x, p: ca_cfar_1d(x, p["num_train"], p["num_guard"], p["pfa"]),
            "go_cfar_1d": lambda x, p: go_cfar_1d(x, p["num_train"], p["num_guard"], p["pfa"]),
            "so_cfar_1d": lambda x, p: so_cfar_1d(x, p["num_train"], p["num_guard"], p["pfa"]),
            "os_cfar_1d": lambda x, p: os_cfar_1d(x, p["num_train"], p["num_guard"], p["rho"], p["alpha"])}


def cfar_1d(kind, x, params):
    """detectors/detector_registry.py:15-27 keys -> detection index list."""
    return _CFAR_1D[kind](x, params)[2]


def _rows_to_dets(mag, rows, vel_kind, vel_params):
    dets = [(int(r), int(d)) for r in rows for d in cfar_1d(vel_kind, mag[r, :], vel_params)]
    return np.array(dets, dtype=int) if dets else np.empty((0, 2), dtype=int)


def rd_detect_sequential(cube, rng_kind, rng_params, vel_kind, vel_params):
    """RangeDopplerDetectorSequential.process -- range_doppler_detector_sequential.py:72-107: 1-D CFAR on the chirp-0
    range profile, then 1-D CFAR along Doppler on |RD| of antenna 0 in every detected range row."""
    rows = cfar_1d(rng_kind, range_profile(cube, 0), rng_params)
    return _rows_to_dets(np.abs(range_doppler(cube)[0]), rows, vel_kind, vel_params)


def find_peaks_db(resp_db, bins, max_peaks=3, threshold_db=20):
    """RangeProcessor.find_peaks -- processors/range_resp.py:104-149 (scipy.signal.find_peaks, prominence 6 dB; peaks
    within threshold_db of the strongest; strongest first; at most max_peaks)."""
    from scipy.signal import find_peaks
    peaks, _ = find_peaks(resp_db, prominence=6)
    if len(peaks) == 0:
        return np.array([]), np.array([])
    vals = resp_db[peaks]
    keep = vals >= (np.max(vals) - threshold_db)
    peaks, vals = peaks[keep], vals[keep]
    top = peaks[np.argsort(vals)[::-1]][:max_peaks]
    return bins[top], resp_db[top]


class Altimeter:
    """processors/altimeter.py:6-140 -- STATEFUL (last-altitude gate): one instance per frame sequence."""

    def __init__(self, sc, min_altitude_m, zoom_search_region_m, altitude_search_limit_m, range_bias=0.0, **_):
        self.sc = sc
        self.range_bins = np.arange(start=0, step=sc["range_res_m"], stop=sc["range_max_m"] - sc["range_res_m"] / 2)
        self.min_altitude_m = float(min_altitude_m)                       # :23-27
        self.zoom_search_region_m = float(zoom_search_region_m)
        self.altitude_search_limit_m = float(altitude_search_limit_m)
        self.range_bias = float(range_bias)
        self.current_altitude_measured_m = self.min_altitude_m            # :34-35
        self.current_altitude_corrected_m = self.min_altitude_m

    def reset(self):                                                      # :37-40 (the corrected value is NOT reset)
        self.current_altitude_measured_m = self.min_altitude_m

    def find_ground_peak(self, peaks_m):                                  # :42-65
        if peaks_m.size > 0:
            valid = peaks_m[(peaks_m >= self.min_altitude_m) &
                            (np.abs(peaks_m - self.current_altitude_measured_m) <= self.altitude_search_limit_m)]
            if valid.size > 0:
                return np.min(valid)
        return -1.0

    def process(self, cube, precise_est_enabled=True, **_):               # :104-140
        coarse = range_profile(cube, 0)
        peaks, _ = find_peaks_db(20 * np.log10(coarse), self.range_bins, max_peaks=3)
        if peaks.size == 0:
            return self.current_altitude_corrected_m
        ground = self.find_ground_peak(peaks)
        if ground < 0:
            return self.current_altitude_corrected_m
        if not precise_est_enabled:
            self.current_altitude_measured_m = ground
            self.current_altitude_corrected_m = ground + self.range_bias
            return self.current_altitude_corrected_m
        lo = max(1e-6, ground - self.zoom_search_region_m)                # :82-88
        hi = min(np.max(self.range_bins) - 1e-6, ground + self.zoom_search_region_m)
        zoom, zbins = range_zoom(cube, self.sc, lo, hi, 0)
        zp, _ = find_peaks_db(20 * np.log10(zoom), zbins, max_peaks=2)
        refined = self.find_ground_peak(zp) if zp.size > 0 else -1.0
        if refined > 0:
            self.current_altitude_measured_m = refined
            self.current_altitude_corrected_m = refined + self.range_bias
        return self.current_altitude_corrected_m


def rd_ground_gate(range_bins, altitude_m):
    """Range rows between the altitude and the 60-degree slant range -- range_doppler_ground_detector.py:91-107."""
    lo = int(np.argmin(np.abs(range_bins - altitude_m)))
    max_rng = min(np.max(range_bins), altitude_m / np.cos(np.deg2rad(60)))
    hi = int(np.argmin(np.abs(range_bins - max_rng)))
    return np.array([lo]) if hi == lo else np.arange(lo, hi + 1)


def rd_detect_ground(cube, altimeter, sc, vel_kind, vel_params, altimeter_params):
    """RangeDopplerGroundDetector.process -- range_doppler_ground_detector.py:72-127 (the detector's own range_bins are
    RangeDopplerProcessor's: range_doppler_resp.py:33-47)."""
    altitude = altimeter.process(cube, **altimeter_params)
    rows = rd_ground_gate(rd_bins(sc)[0], altitude)
    return _rows_to_dets(np.abs(range_doppler(cube)[0]), rows, vel_kind, vel_params), altitude


# --------------------------------------------------------------------------
# a15/a16  PointCloudGenerator -- processors/point_cloud_generator.py:108-248
# --------------------------------------------------------------------------
def angle_argmax(raw, r_idx, v_idx, ant_idxs, num_angle_bins=64, shift=True):
    """Zero-padded FFT over the antenna subset at each detection; returns (argmax idx, |resp| (N,A))."""
    ant = np.asarray(ant_idxs, dtype=int)
    n = len(r_idx)
    batch = raw[ant][:, r_idx, v_idx].T             # (N, n_ant)  :170,177
    pad = np.zeros((n, num_angle_bins), dtype=complex)
    pad[:, :len(ant)] = batch
    f = np.fft.fft(pad, axis=1)
    resp = np.abs(np.fft.fftshift(f, axes=1)) if shift else np.abs(f)
    return np.argmax(resp, axis=1), resp


def point_cloud(cube, sc, az_idxs, el_idxs, num_train=(4, 4), num_guard=(2, 2), pfa=1e-5,
                num_angle_bins=64, shift_az=True, shift_el=False):
    """PointCloudGenerator.process with a RangeDopplerDetector2D/ca_cfar_2d detector -> (N,4) f64."""
    raw, mag, dets, _, _ = rd_detect_2d(cube, num_train, num_guard, pfa)
    if dets.shape[0] == 0:
        return np.empty((0, 4)), dets, None, None
    rbins, vbins = rd_bins(sc)
    r_idx, v_idx = dets[:, 0].astype(int), dets[:, 1].astype(int)
    _, abins = angle_tables(num_angle_bins)
    n = len(r_idx)
    az_i = el_i = None
    az = np.zeros(n)
    el = np.zeros(n)
    if len(az_idxs) > 0:
        az_i, _ = angle_argmax(raw, r_idx, v_idx, az_idxs, num_angle_bins, shift_az)
        az = abins[az_i]
    if len(el_idxs) > 0:
        el_i, _ = angle_argmax(raw, r_idx, v_idx, el_idxs, num_angle_bins, shift_el)
        el = abins[el_i]
    rng, vel = rbins[r_idx], vbins[v_idx]
    x = rng * np.cos(el) * np.cos(az)               # :244-246
    y = rng * np.cos(el) * np.sin(az)
    z = rng * np.sin(el)
    return np.column_stack((x, y, z, vel)), dets, az_i, el_i


# --------------------------------------------------------------------------
# a18  delay-and-sum (Bartlett) beamformer --
#      processors/simple_synthetic_array_beamformer_processor_multiFrame.py:474-585
# --------------------------------------------------------------------------
def steering_dirs(az_bins_rad, el_bins_rad):
    """_compute_beam_stearing_vectors :474-488 -> d[3, n_az, n_el]."""
    th, ph = np.meshgrid(az_bins_rad, el_bins_rad, indexing="ij")
    return np.array([np.cos(th) * np.cos(ph), np.sin(th) * np.cos(ph), np.sin(ph)])


def bartlett_response(X_se, P_3e, d, lambda_m):
    """compute_synthetic_response :543-585: X [S,E] c128, P [3,E] f64, d [3,nAz,nEl] -> [S,nAz,nEl] c128.

    Y[:,a,e] = FFT_S( hann(S) * sum_E X[:,E]*hamming(E)*exp(j 2 pi d.P / lambda) ).
    """
    S, E = X_se.shape
    Xw = X_se * np.hamming(E).reshape(1, -1)
    out = np.zeros((S, d.shape[1], d.shape[2]), dtype=complex)
    win = np.hanning(S)
    for a in range(d.shape[1]):
        for e in range(d.shape[2]):
            shifts = np.exp(1j * 2 * np.pi * (d[:, a, e] @ P_3e) / lambda_m)
            out[:, a, e] = np.fft.fft(np.sum(Xw * shifts.reshape(1, -1), axis=1) * win)
    return out


# --------------------------------------------------------------------------
# config 4: Capon/MVDR -- NO UPSTREAM ORACLE: parity unpinned.
# Definition owned by this build:
#   R_r = (1/K) X_r X_r^H + delta*tr(R_r)/V * I,   P(r,theta) = 1 / Re(a^H R_r^-1 a),
#   a_v(theta) = exp(-j pi v sin(theta)).
# --------------------------------------------------------------------------
def capon_steering(num_elements, thetas_rad):
    v = np.arange(num_elements)[:, None]
    return np.exp(-1j * np.pi * v * np.sin(np.asarray(thetas_rad))[None, :])     # [V, T]


def capon_spectrum(X_vrk, thetas_rad, delta=1e-3):
    """X [V, R, K] complex (K snapshots per range bin) -> P [R, T] f64."""
    V, R, K = X_vrk.shape
    A = capon_steering(V, thetas_rad)
    P = np.empty((R, A.shape[1]))
    for r in range(R):
        Xr = X_vrk[:, r, :].astype(complex)
        Rm = (Xr @ Xr.conj().T) / K
        Rm = Rm + delta * np.real(np.trace(Rm)) / V * np.eye(V)
        Ri_A = np.linalg.solve(Rm, A)
        P[r] = 1.0 / np.real(np.sum(A.conj() * Ri_A, axis=0))
    return P
