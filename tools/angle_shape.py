#!/usr/bin/env python3
"""Time mmw_angle_fft alone (RD cube resident) on several plane sizes: does the store stream care about row alignment?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import _lib  # noqa: E402

ctx = _lib.Context(0)
L = ctx.lib
A, V = 64, 12
FLAGS = int(os.environ.get("ANGLE_FLAGS", "0"))      # 1 = float32 magnitude output
for bins in [int(x) for x in (sys.argv[1:] or "6300 6400 6272 6304 4410 4416 12700 12800 32768".split())]:
    frames = min(4096, (3 << 30) // (A * bins * 8))
    d_rd, d_out = ctx.alloc(frames * V * bins * 8), ctx.alloc(frames * A * bins * 8)
    _lib.check(L.mmw_synth_cubes(ctx.handle, d_rd.ptr, frames, V, bins, 1, 5, 8, 30.0))
    fn = lambda: _lib.check(L.mmw_angle_fft(ctx.handle, d_rd.ptr, d_out.ptr, frames, V, bins, 1, A, FLAGS))
    fn()
    ctx.sync()
    ctx.timer_start()
    for _ in range(5):
        fn()
    ms = ctx.timer_stop() / 5
    print(f"angle bins={bins} frames={frames}: {1e3 * ms / frames:.3f} us/frame, {(V * 8 + A * (4 if FLAGS & 1 else 8)) * bins * frames / ms / 1e6:.0f} GB/s", flush=True)
    d_rd.free()
    d_out.free()
