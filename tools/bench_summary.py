#!/usr/bin/env python3
"""Headline numbers of a bench.py JSON line:  python tools/bench_summary.py line.json"""
import json
import sys

d = json.load(open(sys.argv[1]))
print("chain:", round(d["value"]), "frames/s, roofline frac", round(d["roofline"]["frac"], 4), "|", d["config"]["device"])
print("parity:", d.get("parity_max_rel_err"))
for k in ("detect", "detect_os"):
    if k in d:
        r = d[k]
        print(k + ":", round(r["value"]), "frames/s =", round(r["hbm_frac_of_8TBs"], 4), "of 8 TB/s;", r["ms_per_step"], "ms/step; stages",
              {a: round(b, 4) for a, b in r.get("kernels_ms_per_step", {}).items()}, "sum/step", round(r.get("kernels_ms_sum_over_ms_per_step", 0), 3))
        print("   parity", r.get("parity"), "dets/frame", r.get("detections_per_frame"), "cpu", (r.get("cpu_baseline") or {}).get("value"))
