#!/usr/bin/env python3
"""The Bartlett contraction at the reference's size for 1 / 2 / 4 / 16 frames (S = E = 256, 64 directions): whole call and the
contraction kernel alone (HIP events), with the 16 x 16-tile kernel (default for a frame or two) and with 32 x 32 tiles only."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
L = ctx.lib
rng = np.random.default_rng(1)
lam = 299792458.0 / 77e9
S, E, T = 256, 256, 64
out = {}
for F in (1, 2, 4, 16):
    X = (rng.standard_normal((F, S, E)) + 1j * rng.standard_normal((F, S, E))).astype(np.complex64)
    d_X = ctx.alloc(X.nbytes); d_X.upload(X)
    d_P = ctx.alloc(F * 3 * E * 8); d_P.upload(rng.uniform(-0.05, 0.05, (F, 3, E)))
    az = np.linspace(-1.2, 1.2, T)
    dirs = np.ascontiguousarray(np.stack([np.cos(az), np.sin(az), np.zeros(T)]))
    d_D = ctx.alloc(dirs.nbytes); d_D.upload(dirs)
    d_Y = ctx.alloc(F * S * T * 8)
    fn = lambda: _lib.check(L.mmw_bartlett(ctx.handle, d_X.ptr, d_P.ptr, d_D.ptr, d_Y.ptr, F, S, E, T, lam))
    for tag, t16 in (("tiles16_default", None), ("tiles32_only", 0), ("tiles16_forced", 1)):
        ctx.set_option("MMW_BARTLETT_TILE16", t16)
        for _ in range(5):
            fn()
        ctx.sync()
        ctx.profile_reset(); ctx.profile_enable(1)
        ctx.timer_start()
        for _ in range(50):
            fn()
        ms = ctx.timer_stop() / 50
        ctx.sync()
        cg, n = ctx.profile_get("cgemm")
        ctx.profile_enable(False)
        out[f"F{F}_{tag}"] = {"call_us": round(1e3 * ms, 2), "contraction_us": round(1e3 * cg / max(n, 1), 2),
                              "contraction_TFLOPs": round(8.0 * F * S * E * T / (cg / max(n, 1) * 1e-3) / 1e12, 1)}
    ctx.set_option("MMW_BARTLETT_TILE16", None)
    for b in (d_X, d_P, d_D, d_Y):
        b.free()
print(json.dumps(out, indent=1))
