#!/usr/bin/env python3
"""Condense the passes of tools/pmc_coexec.sh:  python tools/pmc_coexec_summary.py gpurun_out/prof profiles/r04_coexec.json"""
import glob
import json
import os
import sqlite3
import sys

root, out_path = sys.argv[1], sys.argv[2]
out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES (one pass, program "
               "directly after --); per kernel: sums over its dispatches.  coexec_over_mfma_busy = the fraction of matrix-pipe-busy cycles in "
               "which a vector instruction executed as well; valu_quad_cycles x 4 / mfma_busy = vector work offered per matrix-busy cycle."}
for d in sorted(glob.glob(os.path.join(root, "coexec_*"))):
    if not os.path.isdir(d):
        continue
    tag = os.path.basename(d)[len("coexec_"):]
    rows = {}
    for db_path in glob.glob(os.path.join(d, "**", "*results.db"), recursive=True):
        db = sqlite3.connect(db_path)
        for kernel, counter, n, total in db.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from "
                                                    "counters_collection group by kernel_name, counter_name"):
            name = kernel.split("(")[0].replace("void ", "")
            if not any(k in name for k in ("k_diag_mfma", "k_rd_mixed_ct", "k_bartlett_tile", "k_cgemm_mfma", "k_capon_sweep")):
                continue
            rows.setdefault(name, {})[counter] = total
            rows[name]["dispatches"] = n
    for name, c in rows.items():
        busy, co = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0)
        e = {"dispatches": c.get("dispatches"), "mfma_busy_cycles": busy, "coexec_cycles": co, "valu_quad_cycles": c.get("SQ_ACTIVE_INST_VALU"),
             "sq_busy_cycles": c.get("SQ_BUSY_CYCLES")}
        if busy:
            e["coexec_over_mfma_busy"] = round(co / busy, 4)
            if c.get("SQ_ACTIVE_INST_VALU"):
                e["valu_cycles_over_mfma_busy"] = round(4 * c["SQ_ACTIVE_INST_VALU"] / busy, 4)
        out.setdefault(tag, {})[name] = e
    txt = os.path.join(root, f"coexec_{tag}.txt")
    if os.path.exists(txt):
        lines = [l for l in open(txt).read().splitlines() if l.startswith("kind ")]
        if lines:
            out.setdefault(tag, {})["probe_output"] = lines[-1]
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps(out, indent=1))
