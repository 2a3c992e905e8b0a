#!/usr/bin/env python3
"""Randomised differential run of mmw_chain3d: random shipped cube shape, batch length, output flag and schedule
(events / device-synchronised, forced), each compared with the serial schedule on the same input, back to back in one
process (every call changes the ring layout of the one before).  Prints one line per case and a summary."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mmwave_radar_processing_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=2026)
args = ap.parse_args()
with open(os.path.join(ROOT, "tests", "golden", "cfg_scalars.json")) as f:
    cfgs = json.load(f)
shapes = sorted({(e["expect"]["num_rx"] * e["expect"]["num_tx"], e["expect"]["num_samples"], e["expect"]["loops"]) for e in cfgs.values()})
rng = np.random.default_rng(args.seed)
ctx = _lib.Context(0)
L, h = ctx.lib, ctx.handle
A = 64
worst, bad = 0.0, 0
for case in range(args.cases):
    V, S, C = shapes[rng.integers(len(shapes))]
    F = int(rng.choice([1, 2, 7, 33, 150, 400, int(rng.integers(1, 700))]))
    F = max(1, min(F, (3 << 30) // (A * S * C * 8)))
    flags = int(rng.choice([0, 0, 1]))
    mode = str(rng.choice(["events", "sync", "sync"]))
    esz = 4 if flags & 1 else 8
    d_in, d_ref, d_out = ctx.alloc(F * V * S * C * 8), ctx.alloc(F * A * S * C * esz), ctx.alloc(F * A * S * C * esz)
    _lib.check(L.mmw_synth_cubes(h, d_in.ptr, F, V, S, C, int(rng.integers(1 << 30)), 6, 30.0))
    os.environ["MMW_CHAIN_PIPELINE"] = "0"
    _lib.check(L.mmw_chain3d(h, d_in.ptr, None, d_ref.ptr, F, V, S, C, A, flags))
    os.environ["MMW_CHAIN_PIPELINE"] = "1"
    os.environ["MMW_CHAIN_MODE"] = mode
    for rep in range(2):                                    # twice: the second call starts with the ring in use
        _lib.check(L.mmw_chain3d(h, d_in.ptr, None, d_out.ptr, F, V, S, C, A, flags))
    dt = np.float32 if flags & 1 else np.complex64
    ref, got = d_ref.download((F, A, S, C), dt), d_out.download((F, A, S, C), dt)
    dev = float(np.max(np.abs(got - ref)) / max(float(np.max(np.abs(ref))), 1e-30))
    worst = max(worst, dev)
    ok = dev <= 1e-6
    bad += not ok
    print(f"case {case:3d}: {V}x{S}x{C} F={F:4d} flags={flags} {mode:6s} deviation {dev:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
    for b in (d_in, d_ref, d_out):
        b.free()
ctx.sync()
print(f"{args.cases} cases, worst deviation from the serial schedule {worst:.2e} of the peak, {bad} mismatches")
sys.exit(1 if bad else 0)
