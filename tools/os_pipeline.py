#!/usr/bin/env python3
"""OS-CFAR point-cloud pipeline on resident cubes -- mmw_detect_batch (float64 CFAR plane, GUI parameters (5,5)/(3,2), rho 0.7,
alpha 2: ~470 mostly noise-level detections per 256 x 128 frame) + mmw_angle_argmax_exact for both antenna lists -- in
us/frame (HIP events): with the worst-case argmax bound through the dense float64 refinement (mmw_cells64.h), with the
worst-case bound through the direct sums only, and with round 3's empirical eighth of the bound.

    python tools/os_pipeline.py [--frames 256] [--reps 5]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmwave_radar_processing_amd import _lib  # noqa: E402

V, S, C, A = 12, 256, 128, 64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only-default", action="store_true", help="the default variant only (for a kernel trace)")
    args = ap.parse_args()
    F, cap = args.frames, 2048
    ctx = _lib.Context(0)
    L = ctx.lib
    n = S * C
    d_in, d_rd = ctx.alloc(F * V * n * 8), ctx.alloc(F * V * n * 8)
    _lib.check(L.mmw_synth_cubes(ctx.handle, d_in.ptr, F, V, S, C, 880000, 8, 30.0))
    d_mag, d_mask, d_l1 = ctx.alloc(F * n * 8), ctx.alloc(F * n), ctx.alloc(F * V * 4)
    d_dets, d_cnt = ctx.alloc(F * cap * 8), ctx.alloc(F * 4)
    d_az, d_el = ctx.alloc(F * cap * 4), ctx.alloc(F * cap * 4)
    az, n_az = _lib.int_array(range(8))
    el, n_el = _lib.int_array(range(8, 12))
    k_rank = max(1, min(int(0.7 * 220), 220))
    refined = [_lib.C.c_int(0), _lib.C.c_int(0)]

    def run(count=False):
        _lib.check(L.mmw_detect_batch(ctx.handle, d_in.ptr, d_rd.ptr, d_mag.ptr, d_mask.ptr, d_dets.ptr, d_cnt.ptr, d_l1.ptr, F, V, S, C,
                                      1, 5, 5, 3, 2, 2.0, k_rank, cap))
        for i, (ant, na, d_idx, shift) in enumerate(((az, n_az, d_az, 1), (el, n_el, d_el, 0))):
            _lib.check(L.mmw_angle_argmax_exact(ctx.handle, d_in.ptr, d_l1.ptr, d_rd.ptr, d_dets.ptr, d_cnt.ptr, d_idx.ptr, F, V, S, C,
                                                cap, ant, na, A, shift, _lib.C.byref(refined[i]) if count else None))

    out = {"frames": F}
    ref = None
    for tag, opts in (("worst_case_bound_dense", {}),
                      ("worst_case_bound_direct_sums", {"MMW_ARGMAX_DENSE_MIN": 1 << 30}),
                      ("eighth_of_the_bound_direct_sums", {"MMW_ARGMAX_DENSE_MIN": 1 << 30, "MMW_ARGMAX_BOUND_DIV": 8}))[:1 if args.only_default else 3]:
        for k in ("MMW_ARGMAX_DENSE_MIN", "MMW_ARGMAX_BOUND_DIV"):
            ctx.set_option(k, opts.get(k))
        run(count=True)
        ctx.sync()
        ctx.profile_reset()
        ctx.profile_enable(1)
        run()
        ctx.sync()
        stages = {}
        for fam in ("rd", "rd64", "cfar", "compact", "plane_l1", "argmax"):
            t, k = ctx.profile_get(fam)
            if k:
                stages[fam + "_us_per_frame"] = round(1e3 * t / F, 2)
        ctx.profile_enable(False)
        ctx.timer_start()
        for _ in range(args.reps):
            run()
        ms = ctx.timer_stop() / args.reps
        idx = (d_az.download((F, cap), np.int32), d_el.download((F, cap), np.int32))
        cnt = d_cnt.download((F,), np.int32)
        if ref is None:
            ref = idx
        same = all(np.array_equal(x[f, :cnt[f]], y[f, :cnt[f]]) for x, y in zip(idx, ref) for f in range(F))
        out[tag] = {"us_per_frame": round(1e3 * ms / F, 2), "evaluations_refined": [refined[0].value, refined[1].value],
                    "detections": int(cnt.sum()), "indices_equal_to_first_variant": bool(same), "stages": stages}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
