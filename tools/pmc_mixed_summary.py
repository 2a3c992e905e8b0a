#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes of tools/pmc_mixed.sh into profiles/rNN_mixed_pmc.json.

    [SHAPES="12,254,50 12,63,127"] python tools/pmc_mixed_summary.py gpurun_out/prof profiles/rNN_mixed_pmc.json
"""
import glob
import json
import os
import sqlite3
import sys

root, out_path = sys.argv[1], sys.argv[2]
FRAMES, V, REPS_COUNTED = 2048, 12, None


def counters(tag, kind):
    res = {}
    for db_path in glob.glob(os.path.join(root, f"mix_{tag}_{kind}", "**", "*results.db"), recursive=True):
        db = sqlite3.connect(db_path)
        for kernel, counter, n, total in db.execute(
                "select kernel_name, counter_name, count(distinct dispatch_id), sum(value) from counters_collection "
                "group by kernel_name, counter_name"):
            if "k_rd_mixed_ct" in kernel:
                res[counter] = (total, n, kernel.split("(")[0].replace("void ", ""))
    return res


out = {"note": "rocprofv3 --pmc passes (tools/pmc_mixed.sh: FETCH_SIZE, WRITE_SIZE, SQ counters, MFMA busy, GRBM_GUI_ACTIVE in "
               "separate runs, program directly after --) around tools/rd_prof.py: launches of mmw_range_doppler on 2048 frames "
               "of 12 planes; values are per launch.  FETCH_SIZE doubled per the gfx950 note (MI355X_MICROARCH.md, HBM).  "
               "SQ_* counters are sums over all waves; SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES count quad-cycles."}
SHAPES = os.environ.get("SHAPES", "12,63,100 12,254,50").split()
for tag, (S, C) in ((sh.replace(",", "x"), tuple(int(x) for x in sh.split(",")[1:])) for sh in SHAPES):
    planes = FRAMES * V
    cells = planes * S * C
    alg = 2 * cells * 8
    f, w, sq, g, mf = (counters(tag, k) for k in ("fetch", "write", "sq", "grbm", "mfma"))
    per = lambda d, k: d[k][0] / d[k][1]
    e = {"kernel": (f.get("FETCH_SIZE") or (0, 1, "?"))[2], "planes_per_launch": planes, "algorithmic_bytes_per_launch": alg}
    if "FETCH_SIZE" in f and "WRITE_SIZE" in w:
        e["fetch_bytes_corrected"] = 2 * 1024 * per(f, "FETCH_SIZE") if per(f, "FETCH_SIZE") < alg / 512 else 2 * per(f, "FETCH_SIZE")
        e["write_bytes"] = 1024 * per(w, "WRITE_SIZE") if per(w, "WRITE_SIZE") < alg / 512 else per(w, "WRITE_SIZE")
        e["traffic_over_algorithmic"] = (e["fetch_bytes_corrected"] + e["write_bytes"]) / alg
    if sq:
        e["lds_bank_conflict_fraction_of_lds_cycles"] = per(sq, "SQ_LDS_BANK_CONFLICT") / max(1.0, per(sq, "SQ_LDS_IDX_ACTIVE"))
        e["valu_instructions_per_cell"] = per(sq, "SQ_INSTS_VALU") * 64 / cells
        e["lds_instructions_per_cell"] = per(sq, "SQ_INSTS_LDS") * 64 / cells
        e["wait_any_fraction_of_wave_time"] = per(sq, "SQ_WAIT_ANY") / max(1.0, per(sq, "SQ_WAVE_CYCLES"))
        e["raw_sq"] = {k: per(sq, k) for k in sq}
    if g:
        cyc = per(g, "GRBM_GUI_ACTIVE") / 8            # summed over the 8 XCDs
        e["kernel_cycles (GRBM_GUI_ACTIVE / 8)"] = cyc
        simd_cycles = cyc * 1024                        # 256 CUs x 4 SIMDs
        if sq:
            e["valu_busy_fraction_of_simd_time"] = 4 * per(sq, "SQ_ACTIVE_INST_VALU") / simd_cycles
        if mf:
            e["mfma_busy_fraction_of_simd_time"] = per(mf, "SQ_VALU_MFMA_BUSY_CYCLES") / simd_cycles
    if mf:
        e["raw_mfma"] = {k: per(mf, k) for k in mf}
    out[tag] = e
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if not kk.startswith("raw")} for k, v in out.items() if k != "note"}, indent=1))
