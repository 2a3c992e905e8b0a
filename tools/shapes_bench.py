#!/usr/bin/env python3
"""Range-Doppler and 3-D chain timings for cube shapes other than the headline one (generic kernels)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
L = ctx.lib
out = {}
# power-of-two planes, then the (samples, loops) planes of the cfg files the reference ships
SHAPES = ((12, 64, 64), (12, 128, 128), (12, 256, 128), (12, 512, 64), (12, 128, 256), (12, 256, 256), (12, 512, 128),
          # (virtual antennas, samples, loops) of the 25 cfg files the reference ships (tests/golden/cfg_scalars.json)
          (8, 63, 100), (8, 63, 115), (8, 63, 127), (8, 64, 40), (8, 64, 64), (8, 90, 80), (8, 90, 100), (8, 127, 32),
          (8, 130, 50), (8, 200, 40), (8, 254, 50), (8, 512, 8), (8, 512, 32), (12, 63, 70), (12, 63, 100), (12, 70, 40),
          (12, 100, 30), (12, 100, 100), (12, 120, 126),
          # same planes with 12 antennas, as in round 1's table
          (12, 90, 100), (12, 200, 40), (12, 254, 50), (12, 63, 127), (12, 130, 50), (12, 63, 115))
for (V, S, C) in SHAPES:
    A = 64
    frames = max(8, min(2048, (4 << 30) // (V * S * C * 8 * 8)))      # <= 4 GiB of 3-D output, >= ~20 waves of workgroups
    n = V * S * C * 8
    d_in, d_rd, d_out = ctx.alloc(frames * n), ctx.alloc(frames * n), ctx.alloc(frames * A * S * C * 8)
    _lib.check(L.mmw_synth_cubes(ctx.handle, d_in.ptr, frames, V, S, C, 5, 8, 30.0))
    res = {}
    for name, fn, moved in (
            ("rd", lambda: _lib.check(L.mmw_range_doppler(ctx.handle, d_in.ptr, d_rd.ptr, None, frames, V, S, C)), 2 * n),
            ("chain3d", lambda: _lib.check(L.mmw_chain3d(ctx.handle, d_in.ptr, None, d_out.ptr, frames, V, S, C, A, 0)),
             n + A * S * C * 8)):
        fn(); ctx.sync(); ctx.timer_start()
        for _ in range(5):
            fn()
        ms = ctx.timer_stop() / 5
        res[name] = {"us_per_frame": round(1e3 * ms / frames, 3), "GBs": round(frames * moved / ms / 1e6)}
    if (S & (S - 1)) or (C & (C - 1)):       # same call on the generic two-kernel path, for comparison
        os.environ["MMW_NO_MIXED_RD"] = "1"
        fn = lambda: _lib.check(L.mmw_range_doppler(ctx.handle, d_in.ptr, d_rd.ptr, None, frames, V, S, C))
        fn(); ctx.sync(); ctx.timer_start()
        for _ in range(5):
            fn()
        ms = ctx.timer_stop() / 5
        res["rd_generic_path"] = {"us_per_frame": round(1e3 * ms / frames, 3), "GBs": round(frames * 2 * n / ms / 1e6)}
        del os.environ["MMW_NO_MIXED_RD"]
    out[f"{V}x{S}x{C}"] = dict(frames=frames, **res)
    for b in (d_in, d_rd, d_out):
        b.free()
print(json.dumps(out))
