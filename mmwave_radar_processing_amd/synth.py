"""Synthetic IWR1843-style inputs: TI ``.cfg`` text and raw ADC cubes.

Input generation only -- no signal processing happens here.  The recipe is
the one SURVEY.md section 8(d) fixes for the headline workload: K point
targets plus complex Gaussian noise, rounded to integers (16-bit-ADC-like)
and stored complex64 in the reference's cube layout ``[virtRx, sample,
chirp]``, C-order, chirp fastest (reference:
mmwave_radar_processing/processors/_processor.py:58).
"""
from __future__ import annotations

import numpy as np

# 4 Rx x 3 Tx (TDM), 256 samples, 128 loops -> virtual cube (12, 256, 128).
# Field order per TI mmWave SDK; parsed by config_managers.ConfigManager.
SYNTH_CFG_256x128x12 = """\
channelCfg 15 7 0
adcCfg 2 1
profileCfg 0 77 7 7 60 0 0 30 1 256 5000 0 0 30
chirpCfg 0 0 0 0 0 0 0 1
chirpCfg 1 1 0 0 0 0 0 2
chirpCfg 2 2 0 0 0 0 0 4
frameCfg 0 2 128 0 100 1 0
"""


def synth_cfg_text(num_samples: int = 256, num_loops: int = 128, num_tx: int = 3,
                   rx_mask: int = 15, sample_rate_ksps: int = 5000,
                   slope_mhz_us: float = 30.0, start_freq_ghz: float = 77.0) -> str:
    """Text of a minimal TI cfg producing ``(popcount(rx_mask)*num_tx, num_samples, num_loops)``."""
    tx_mask = (1 << num_tx) - 1
    lines = [
        f"channelCfg {rx_mask} {tx_mask} 0",
        "adcCfg 2 1",
        f"profileCfg 0 {start_freq_ghz:g} 7 7 60 0 0 {slope_mhz_us:g} 1 {num_samples} {sample_rate_ksps} 0 0 30",
    ]
    for i in range(num_tx):
        lines.append(f"chirpCfg {i} {i} 0 0 0 0 0 {1 << i}")
    lines.append(f"frameCfg 0 {num_tx - 1} {num_loops} 0 100 1 0")
    return "\n".join(lines) + "\n"


def synth_cube(seed: int, shape=(12, 256, 128), num_targets: int = 8,
               noise_sigma: float = 30.0) -> np.ndarray:
    """One virtual-array ADC cube ``[V, S, C]`` complex64 with integer-valued I/Q.

    ``default_rng(seed)`` (PCG64) is stable across numpy versions, so tests
    and the golden generator regenerate identical inputs from the seed.
    """
    V, S, C = shape
    rng = np.random.default_rng(seed)
    f_r = rng.uniform(0.02, 0.45, num_targets)
    f_d = rng.uniform(-0.45, 0.45, num_targets)
    f_a = rng.uniform(-0.4, 0.4, num_targets)
    amp = rng.uniform(20.0, 400.0, num_targets)
    phi = rng.uniform(0.0, 2.0 * np.pi, num_targets)
    v = np.arange(V)[:, None, None]
    n = np.arange(S)[None, :, None]
    m = np.arange(C)[None, None, :]
    x = np.zeros(shape, dtype=np.complex128)
    for k in range(num_targets):
        x += amp[k] * np.exp(1j * (2.0 * np.pi * (f_r[k] * n + f_d[k] * m + f_a[k] * v) + phi[k]))
    x += noise_sigma * (rng.standard_normal(shape) + 1j * rng.standard_normal(shape))
    return (np.rint(x.real) + 1j * np.rint(x.imag)).astype(np.complex64)


def synth_raw_cube(seed: int, num_rx: int = 4, num_tx: int = 3, num_samples: int = 256,
                   num_loops: int = 128) -> np.ndarray:
    """Raw TDM-interleaved cube ``[num_rx, S, num_tx*loops]`` (input of VirtualArrayReformatter).

    Built by interleaving a virtual cube so that de-interleaving it gives
    ``synth_cube(seed, (num_rx*num_tx, S, loops))`` back.
    """
    virt = synth_cube(seed, (num_rx * num_tx, num_samples, num_loops))
    raw = np.empty((num_rx, num_samples, num_tx * num_loops), dtype=np.complex64)
    for t in range(num_tx):
        raw[:, :, t::num_tx] = virt[t * num_rx:(t + 1) * num_rx]
    return raw


def synth_ground_sequence(seed: int, n_frames: int = 5, shape=(12, 256, 128), range_res_m: float = 0.0975887,
                          altitude0_m: float = 1.02, climb_m: float = 0.07, ground_amp: float = 350.0,
                          num_clutter: int = 3, noise_sigma: float = 30.0) -> np.ndarray:
    """``[n_frames, V, S, C]`` cubes of a platform over flat ground: one strong return whose range grows by ``climb_m``
    per frame (what ``Altimeter`` tracks), spread over a few Doppler lines, plus ``num_clutter`` weaker point targets
    farther out and complex Gaussian noise; integer-valued complex64 like ``synth_cube``."""
    V, S, C = shape
    rng = np.random.default_rng(seed)
    v = np.arange(V)[:, None, None]
    n = np.arange(S)[None, :, None]
    m = np.arange(C)[None, None, :]
    cubes = np.empty((n_frames,) + tuple(shape), dtype=np.complex64)
    for f in range(n_frames):
        x = np.zeros(shape, dtype=np.complex128)
        alt = altitude0_m + climb_m * f
        for dv, a in ((-0.06, 0.5), (0.0, 1.0), (0.05, 0.6)):          # ground patch: a few Doppler lines
            fr = alt / range_res_m / S
            x += a * ground_amp * np.exp(1j * (2.0 * np.pi * (fr * n + dv * m + 0.02 * v) + rng.uniform(0, 2 * np.pi)))
        for _ in range(num_clutter):
            fr = rng.uniform(2.5, 9.0) / range_res_m / S
            x += rng.uniform(30.0, 120.0) * np.exp(1j * (2.0 * np.pi * (fr * n + rng.uniform(-0.4, 0.4) * m +
                                                                        rng.uniform(-0.3, 0.3) * v) + rng.uniform(0, 2 * np.pi)))
        x += noise_sigma * (rng.standard_normal(shape) + 1j * rng.standard_normal(shape))
        cubes[f] = (np.rint(x.real) + 1j * np.rint(x.imag)).astype(np.complex64)
    return cubes
