#!/usr/bin/env python3
"""What the in-kernel angle argmax costs the screening launch of mmw_detect_points: the pipeline with both antenna lists, with
none, for several MMW_DETECT_TAIL_CUS (CUs the range-Doppler kernel leaves to the previous call's deferred tail).

    python tools/detect_probe.py [--frames 1250] [--reps 10]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import _lib  # noqa: E402

V, S, C, A = 12, 256, 128, 64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1250)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--tail-cus", default="40,32,24")
    ap.add_argument("--cap", type=int, default=1024)
    ap.add_argument("--settle", action="store_true")
    ap.add_argument("--ballast-gb", type=float, default=0.0, help="allocate this much before the pipeline's buffers (bench.py holds 21 GB of chain output)")
    ap.add_argument("--opts", default="", help="extra context options NAME=VALUE[,NAME=VALUE...] tried one at a time")
    args = ap.parse_args()
    F, reps = args.frames, args.reps
    ctx = _lib.Context(0)
    L = ctx.lib
    cube_b, cap = V * S * C * 8, args.cap
    d_in = ctx.alloc(F * cube_b)
    ballast = ctx.alloc(int(args.ballast_gb * 2**30)) if args.ballast_gb > 0 else None
    d_rd = ctx.alloc(F * cube_b)
    _lib.check(L.mmw_synth_cubes(ctx.handle, d_in.ptr, F, V, S, C, 99, 8, 30.0))
    d_dets, d_cnt = ctx.alloc(F * cap * 8), ctx.alloc(F * 4)
    d_az, d_el, d_l1 = ctx.alloc(F * cap * 4), ctx.alloc(F * cap * 4), ctx.alloc(F * V * 4)
    d_m32 = ctx.alloc(F * S * C * 4)
    az8, n_az = _lib.int_array(range(8))
    el4, n_el = _lib.int_array(range(8, 12))
    alpha = 144 * (1e-5 ** (-1.0 / 144) - 1.0)

    def run(lists):
        _lib.check(L.mmw_detect_points(ctx.handle, d_in.ptr, d_rd.ptr, d_l1.ptr, d_m32.ptr, d_dets.ptr, d_cnt.ptr,
                                       d_az.ptr if lists else None, d_el.ptr if lists else None, F, V, S, C, 0, 4, 4, 2, 2, alpha, 0, cap,
                                       az8, n_az if lists else 0, 1, el4, n_el if lists else 0, 0, A, None))

    def measure(tag, lists):
        run(lists)
        ctx.sync()
        ctx.profile_reset()
        ctx.profile_enable(1)
        ctx.timer_start()
        for _ in range(reps):
            run(lists)
        total = ctx.timer_stop() / reps
        ctx.sync()
        out = {"case": tag, "total_ms": round(total, 4), "frac_of_8TBs": round(F * 6422528 / (total * 1e-3) / 8e12, 4)}
        for fam in ("rd", "detect", "detect_exact", "argmax_tail", "argmax_refine"):
            ms, n = ctx.profile_get(fam)
            if n:
                out[fam + "_ms"] = round(ms / n, 4)
        ctx.profile_enable(False)
        print(json.dumps(out), flush=True)

    if args.settle:        # how many calls until the rate settles: the same case over and over, 5 calls each
        for i in range(12):
            measure(f"lists, defer=1, block {i} of {reps} calls", True)
        return
    for defer in (1, 0):
        ctx.set_option("MMW_DETECT_DEFER_TAIL", defer)
        measure(f"lists, defer={defer}", True)
        measure(f"no lists, defer={defer}", False)
    ctx.set_option("MMW_DETECT_DEFER_TAIL", 1)
    for k in (int(x) for x in args.tail_cus.split(",")):
        ctx.set_option("MMW_DETECT_TAIL_CUS", k)
        measure(f"lists, defer=1, tail_cus={k}", True)
    ctx.set_option("MMW_DETECT_TAIL_CUS", None)
    for kv in [x for x in args.opts.split(",") if x]:
        name, val = kv.split("=")
        ctx.set_option(name, int(val))
        measure(f"lists, defer=1, {kv}", True)
        ctx.set_option("MMW_DETECT_DEFER_TAIL", 0)
        measure(f"lists, defer=0, {kv}", True)
        ctx.set_option("MMW_DETECT_DEFER_TAIL", 1)
        ctx.set_option(name, None)


if __name__ == "__main__":
    main()
