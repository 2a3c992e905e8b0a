#!/usr/bin/env python3
"""Schedules of the configs[2] detection pipeline (mmw_detect_points): the serial one against the overlapped one (range-Doppler
producer || screening consumer on disjoint CU sets) for several CU splits, with and without the tail consumer.  Stage
times are HIP-event spans on the queue each stage runs on (under the overlapped schedule they overlap: the sum exceeds
the total).

    python tools/detect_cus.py [--frames 1250] [--reps 10] [--splits 32,64,96]
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
    ap.add_argument("--splits", default="32,64,96")
    ap.add_argument("--naps", default="4")
    ap.add_argument("--mag32", action="store_true", help="also ask for the float32 |RD| plane of antenna 0 (as bench.py does)")
    ap.add_argument("--clocks", action="store_true", help="phase clocks of consumer workgroup 0 (stderr) instead of timings")
    args = ap.parse_args()
    F, reps = args.frames, args.reps
    ctx = _lib.Context(0)
    L = ctx.lib
    cube_b, cap = V * S * C * 8, 1024
    d_in, d_rd = ctx.alloc(F * cube_b), ctx.alloc(F * cube_b)
    _lib.check(L.mmw_synth_cubes(ctx.handle, d_in.ptr, F, V, S, C, 99, 8, 30.0))
    d_dets, d_cnt = ctx.alloc(F * cap * 8), ctx.alloc(F * 4)
    d_az, d_el, d_l1 = ctx.alloc(F * cap * 4), ctx.alloc(F * cap * 4), ctx.alloc(F * V * 4)
    az8, n_az = _lib.int_array(range(8))
    el4, n_el = _lib.int_array(range(8, 12))
    alpha = 144 * (1e-5 ** (-1.0 / 144) - 1.0)
    stats = (_lib.C.c_int * 5)()
    d_m32 = ctx.alloc(F * S * C * 4) if args.mag32 else None

    def run(st=None):
        _lib.check(L.mmw_detect_points(ctx.handle, d_in.ptr, d_rd.ptr, d_l1.ptr, d_m32.ptr if d_m32 else None, d_dets.ptr, d_cnt.ptr, d_az.ptr, d_el.ptr, F,
                                       V, S, C, 0, 4, 4, 2, 2, alpha, 0, cap, az8, n_az, 1, el4, n_el, 0, A, st))

    def measure(tag):
        run(stats)
        ctx.sync()
        ctx.profile_reset()
        ctx.profile_enable(1)
        ctx.timer_start()
        for _ in range(reps):
            run()
        total = ctx.timer_stop() / reps
        ctx.sync()
        out = {"schedule": tag, "total_ms": round(total, 4), "frames_per_s": round(F / total * 1e3),
               "frac_of_8TBs": round(F * 6.42e6 / (total * 1e-3) / 8e12, 4), "stats": list(stats)}
        for fam in ("rd", "detect", "detect_tail", "detect_exact", "argmax_refine"):
            ms, n = ctx.profile_get(fam)
            if n:
                out[fam + "_ms"] = round(ms / n, 4)
        ctx.profile_enable(False)
        out["detections"] = int(d_cnt.download((F,), "int32").sum())
        print(json.dumps(out), flush=True)

    if args.clocks:
        ctx.set_option("MMW_DETECT_OVERLAP", 1)
        ctx.set_option("MMW_DETECT_TAIL", 0)
        for k in (int(x) for x in args.splits.split(",")):
            ctx.set_option("MMW_DETECT_SCR_CUS", k)
            run()
            ctx.sync()
            ctx.set_option("MMW_PHASE_CLOCKS", 1)
            print(f"scr_cus={k}", file=sys.stderr, flush=True)
            run()
            ctx.sync()
            ctx.set_option("MMW_PHASE_CLOCKS", None)
        return
    ctx.set_option("MMW_DETECT_OVERLAP", 0)
    measure("serial")
    ctx.set_option("MMW_DETECT_OVERLAP", 1)
    for naps in (int(x) for x in args.naps.split(",")):
        ctx.set_option("MMW_DETECT_NAPS", naps)
        for k in (int(x) for x in args.splits.split(",")):
            for tail in (1, 0):
                ctx.set_option("MMW_DETECT_SCR_CUS", k)
                ctx.set_option("MMW_DETECT_TAIL", tail)
                measure(f"overlap scr_cus={k} tail={tail} naps={naps}")


if __name__ == "__main__":
    main()
