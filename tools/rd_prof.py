#!/usr/bin/env python3
"""Minimal driver for profiling one range-Doppler kernel under rocprofv3: N launches of mmw_range_doppler on a batch."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="12,63,100")
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
V, S, C = (int(x) for x in args.shape.split(","))
ctx = _lib.Context(0)
n = V * S * C * 8
d_in, d_rd = ctx.alloc(args.frames * n), ctx.alloc(args.frames * n)
_lib.check(ctx.lib.mmw_synth_cubes(ctx.handle, d_in.ptr, args.frames, V, S, C, 5, 8, 30.0))
for _ in range(args.reps):
    _lib.check(ctx.lib.mmw_range_doppler(ctx.handle, d_in.ptr, d_rd.ptr, None, args.frames, V, S, C))
ctx.sync()
ctx.timer_start()
for _ in range(args.reps):
    _lib.check(ctx.lib.mmw_range_doppler(ctx.handle, d_in.ptr, d_rd.ptr, None, args.frames, V, S, C))
ms = ctx.timer_stop() / args.reps
print(f"rd {args.shape}: {1e3 * ms / args.frames:.3f} us/frame, {2 * n * args.frames / ms / 1e6:.0f} GB/s")
