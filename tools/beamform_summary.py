#!/usr/bin/env python3
"""Build profiles/r02_beamform.json from tools/kbench.py's beamformer entries and the rocprofv3 --pmc database of
tools/beamform_prof.py (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE).

    python tools/beamform_summary.py <kbench.json> <pmc_results.db> <out.json> [<previous r02_beamform.json>]
"""
import json
import sqlite3
import sys

kb = json.load(open(sys.argv[1]))
db = sqlite3.connect(sys.argv[2])
out_path = sys.argv[3]
prev = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else {}
pmc = {}
for kernel, counter, n, total in db.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from "
                                            "counters_collection group by kernel_name, counter_name"):
    for key, label in (("k_capon_batch", "k_capon_batch (32 frames per launch)"),
                       ("k_cgemm_mfma", "k_cgemm_mfma (16 frames x 256 x 256 x 900 per launch)")):
        if key in kernel:
            pmc.setdefault(label, {})[counter] = total / n
for kernel, avg in db.execute("select kernel_name, avg(end - start) from counters_collection group by kernel_name"):
    for key, label in (("k_capon_batch", "k_capon_batch (32 frames per launch)"),
                       ("k_cgemm_mfma", "k_cgemm_mfma (16 frames x 256 x 256 x 900 per launch)")):
        if key in kernel:
            pmc[label]["avg_duration_us_under_pmc"] = avg / 1e3
res = {}
for label, c in pmc.items():
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    res[label] = {"mfma_busy_cycles_per_launch": c["SQ_VALU_MFMA_BUSY_CYCLES"],
                  "kernel_cycles_per_launch (GRBM_GUI_ACTIVE / 8 XCDs)": cyc,
                  "mfma_busy_fraction_of_simd_time": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
                  "SQ_BUSY_CYCLES_per_launch": c["SQ_BUSY_CYCLES"],
                  "avg_duration_us_under_pmc": c.get("avg_duration_us_under_pmc")}
out = {"note": prev.get("note", ""), "mfma_peak_measured_TFLOPs": kb.get("mfma_peak_measured_TFLOPs"),
       "kbench": {k: v for k, v in kb.items() if k.startswith(("bartlett", "capon"))}, "pmc": res,
       "round1_for_comparison": prev.get("round1_for_comparison")}
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({"kbench": out["kbench"], "pmc": res}, indent=1))
