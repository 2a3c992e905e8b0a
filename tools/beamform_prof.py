#!/usr/bin/env python3
"""Minimal driver for profiling the beamformer kernels under rocprofv3 (kernel trace or --pmc passes):
10 launches of the batched Capon kernel (32 frames of 12 x 512 x 128, 181 angles), 10 of the batched Bartlett
contraction at 900 steering directions (16 frames of 256 x 256: the tiled GEMM) and 10 at the reference's own size
(16 frames of 256 x 256, 64 directions: the tile kernel with the steering fused)."""
import ctypes as ct
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import _lib  # noqa: E402


def main():
    ctx = _lib.Context(0)
    L, h = ctx.lib, ctx.handle
    rng = np.random.default_rng(1)
    Fc, V, R, K, T = 32, 12, 512, 128, 181
    Xc = (rng.standard_normal((Fc, V, R, K)) + 1j * rng.standard_normal((Fc, V, R, K))).astype(np.complex64)
    d_Xc, d_Pc = ctx.alloc(Xc.nbytes), ctx.alloc(Fc * R * T * 4)
    d_Xc.upload(Xc)
    th = np.linspace(-1.3, 1.3, T)
    Fb, S, E, Tb = 16, 256, 256, 900
    Xb = (rng.standard_normal((Fb, S, E)) + 1j * rng.standard_normal((Fb, S, E))).astype(np.complex64)
    d_Xb, d_P, d_Y = ctx.alloc(Xb.nbytes), ctx.alloc(Fb * 3 * E * 8), ctx.alloc(Fb * S * Tb * 8)
    d_Xb.upload(Xb)
    d_P.upload(rng.uniform(-0.05, 0.05, (Fb, 3, E)))
    az = np.linspace(-1.2, 1.2, Tb)
    dirs = np.ascontiguousarray(np.stack([np.cos(az), np.sin(az), np.zeros(Tb)]))
    d_D = ctx.alloc(dirs.nbytes)
    d_D.upload(dirs)
    az2 = np.linspace(-1.2, 1.2, 64)
    dirs2 = np.ascontiguousarray(np.stack([np.cos(az2), np.sin(az2), np.zeros(64)]))
    d_D2 = ctx.alloc(dirs2.nbytes)
    d_D2.upload(dirs2)
    for _ in range(10):
        _lib.check(L.mmw_capon(h, d_Xc.ptr, th.ctypes.data_as(ct.POINTER(ct.c_double)), d_Pc.ptr, Fc, V, R, K, T, 1e-3))
        _lib.check(L.mmw_bartlett(h, d_Xb.ptr, d_P.ptr, d_D.ptr, d_Y.ptr, Fb, S, E, Tb, 299792458.0 / 77e9))
        _lib.check(L.mmw_bartlett(h, d_Xb.ptr, d_P.ptr, d_D2.ptr, d_Y.ptr, Fb, S, E, 64, 299792458.0 / 77e9))
    ctx.sync()
    print("beamform_prof done")


if __name__ == "__main__":
    main()
