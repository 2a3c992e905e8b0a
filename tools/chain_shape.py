#!/usr/bin/env python3
"""Time mmw_chain3d (range + Doppler + angle FFT, no RD cube kept) on one cube shape; schedule knobs come from the
environment (MMW_RD_CUS, MMW_CHAIN_CHUNK, MMW_CHAIN_RING, MMW_CHAIN_PIPELINE, MMW_ANGLE_QUEUES)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="12,63,100")
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tag", default="")
ap.add_argument("--rd", action="store_true", help="time mmw_range_doppler (cube out) instead of the chain")
args = ap.parse_args()
V, S, C = (int(x) for x in args.shape.split(","))
A = 64
ctx = _lib.Context(0)
L = ctx.lib
n = V * S * C * 8
d_in, d_out = ctx.alloc(args.frames * n), ctx.alloc(args.frames * A * S * C * 8)
_lib.check(L.mmw_synth_cubes(ctx.handle, d_in.ptr, args.frames, V, S, C, 5, 8, 30.0))
fn = lambda: _lib.check(L.mmw_chain3d(ctx.handle, d_in.ptr, None, d_out.ptr, args.frames, V, S, C, A, 0))
if args.rd:
    fn = lambda: _lib.check(L.mmw_range_doppler(ctx.handle, d_in.ptr, d_out.ptr, None, args.frames, V, S, C))
fn()
ctx.sync()
ctx.timer_start()
for _ in range(args.reps):
    fn()
ms = ctx.timer_stop() / args.reps
moved = 2 * n if args.rd else n + A * S * C * 8
print(f"{args.tag} {'rd' if args.rd else 'chain'} {args.shape}: {1e3 * ms / args.frames:.3f} us/frame, {moved * args.frames / ms / 1e6:.0f} GB/s", flush=True)
