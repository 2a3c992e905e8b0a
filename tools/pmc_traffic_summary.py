#!/usr/bin/env python3
"""Condense the --pmc FETCH_SIZE / WRITE_SIZE databases of tools/pmc_traffic.sh into profiles/rNN_pmc_traffic.json.

    python tools/pmc_traffic_summary.py <dir with chain_* / det_* result dirs> <out.json>
"""
import glob
import json
import sqlite3
import sys

root, out_path = sys.argv[1], sys.argv[2]
F, V, S, C, A = 1250, 12, 256, 128, 64          # bench.py defaults
STEPS = 3


def counters(prefix):
    """kernel short name -> {counter: (launches, sum)}"""
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for path in glob.glob(f"{root}/{prefix}_{c}/**/*results.db", recursive=True):
            db = sqlite3.connect(path)
            for kernel, counter, n, total in db.execute("select kernel_name, counter_name, count(distinct dispatch_id), sum(value) "
                                                        "from counters_collection group by kernel_name, counter_name"):
                short = kernel.split("(")[0].replace("void ", "")
                res.setdefault(short, {})[counter] = (n, total)
    return res


def traffic(entry, alg_total):
    """bytes (FETCH corrected, WRITE) of all launches of one kernel; the counters are KiB"""
    f, w = entry["FETCH_SIZE"][1], entry["WRITE_SIZE"][1]
    fetch = 2 * (1024 * f if f < alg_total / 64 else f)
    write = 1024 * w if w < alg_total / 64 else w
    return fetch, write


frames = F * STEPS
plane = S * C * 8
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, program directly after --) by tools/pmc_traffic.sh. "
               "Chain: `python3 bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-profile --no-detect-record` with MMW_CHAIN_MODE=events "
               "MMW_ANGLE_QUEUES=1 -- counter collection serialises kernel dispatches, which the device-synchronised default schedule "
               "cannot run under (its two launches must overlap), so the traffic is measured on the event-mode kernels k_angle64 / "
               "k_rd_fused_256x128: the same loads and stores per frame as k_angle64_sync / the persistent RD kernel. Detection: "
               "`bench.py --workload detect --steps 3 --warmup 0` with MMW_DETECT_DEFER_TAIL=0 (every call joins its own tail). Counters are KiB summed over all launches; gfx950 "
               "FETCH_SIZE of a wide coalesced read stream reports half the bytes (MI355X_MICROARCH.md, HBM) -> doubled. "
               "Infinity-Cache hits are counted by these L2-side counters. The chain kernels skip the two antennas whose Hann(12) "
               "weight is exactly zero: 10 of 12 planes are read and transformed.",
       "frames_total": frames}
ch = counters("chain")
for key, name, alg, need in (("angle", "k_angle64", (V + A) * plane, (V - 2 + A) * plane),
                             ("rd", "k_rd_fused_256x128", 2 * (V - 2) * plane, 2 * (V - 2) * plane)):
    cand = [k for k in ch if name in k and "FETCH_SIZE" in ch[k] and "WRITE_SIZE" in ch[k]]
    if not cand:
        continue
    k = max(cand, key=lambda x: ch[x]["FETCH_SIZE"][1])
    fetch, write = traffic(ch[k], alg * frames)
    out[key] = {"kernel": k, "launches": ch[k]["FETCH_SIZE"][0], "read_bytes_corrected_per_frame": fetch / frames,
                "write_bytes_per_frame": write / frames, "traffic_bytes_per_frame": (fetch + write) / frames,
                "algorithmic_bytes_per_frame": alg, "bytes_needed_per_frame": need,
                "traffic_over_algorithmic": (fetch + write) / frames / alg, "traffic_over_needed": (fetch + write) / frames / need}
if "angle" in out:
    out["angle_bytes_per_frame"] = out["angle"]["traffic_bytes_per_frame"]
de = counters("det")
det = {}
for k, v in de.items():
    if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
        continue
    if not any(s in k for s in ("k_rd_fused", "k_detect_screen", "k_cfar_cell_exact", "k_detect_insert", "k_angle_argmax_recs", "k_argmax_refine")):
        continue
    fetch, write = traffic(v, 2 * V * plane * frames)
    det[k] = {"launches": v["FETCH_SIZE"][0], "read_bytes_corrected_per_frame": fetch / frames, "write_bytes_per_frame": write / frames}
if det:
    rd = [k for k in det if "k_rd_fused" in k]
    out["detect"] = {"kernels": det, "algorithmic_bytes_per_frame": 2 * V * plane + S * C * 8 + 76 * 2 * 4 * 3,
                     "traffic_bytes_per_frame": sum(v["read_bytes_corrected_per_frame"] + v["write_bytes_per_frame"] for v in det.values())}
    if rd:
        r = det[rd[0]]
        out["detect"]["rd_bytes_per_frame"] = r["read_bytes_corrected_per_frame"] + r["write_bytes_per_frame"]
        out["detect"]["rd_traffic_over_algorithmic"] = out["detect"]["rd_bytes_per_frame"] / (2 * V * plane)
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "note"}, indent=1))
