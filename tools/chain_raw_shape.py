#!/usr/bin/env python3
"""Time mmw_chain3d_raw (raw [F][nrx][S][ntx * C] cubes: de-interleave folded into the range-Doppler stage) on one shape."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="4,3,63,100", help="nrx,ntx,S,C")
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tag", default="")
args = ap.parse_args()
nrx, ntx, S, C = (int(x) for x in args.shape.split(","))
V, A = nrx * ntx, 64
ctx = _lib.Context(0)
L = ctx.lib
n = V * S * C * 8
d_in, d_out = ctx.alloc(args.frames * n), ctx.alloc(args.frames * A * S * C * 8)
_lib.check(L.mmw_synth_cubes(ctx.handle, d_in.ptr, args.frames, nrx, S, ntx * C, 5, 8, 30.0))
fn = lambda: _lib.check(L.mmw_chain3d_raw(ctx.handle, d_in.ptr, None, d_out.ptr, args.frames, nrx, ntx, S, C, A, 0))
fn()
ctx.sync()
ctx.timer_start()
for _ in range(args.reps):
    fn()
ms = ctx.timer_stop() / args.reps
print(f"{args.tag} raw chain {args.shape}: {1e3 * ms / args.frames:.3f} us/frame, {(n + A * S * C * 8) * args.frames / ms / 1e6:.0f} GB/s", flush=True)
