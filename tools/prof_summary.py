#!/usr/bin/env python3
"""Condense rocprofv3 result databases (the default rocpd SQLite output of ROCm 7.2) into the small files kept under
profiles/: a per-kernel duration table (same columns as `--stats` CSV) and per-kernel counter sums.

    python tools/prof_summary.py stats  <results.db> <out.csv>
    python tools/prof_summary.py pmc    <results.db> [<results.db> ...] <out.json>
"""
import csv
import json
import sqlite3
import sys


def stats(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                      "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    with open(out_csv, "w", newline="") as fh:
        w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for name, n, tot, avg, mn, mx in rows:
            w.writerow([name, n, tot, round(avg, 3), round(100.0 * tot / total, 2), mn, mx])


def pmc(db_paths, out_json):
    out = {}
    for path in db_paths:
        db = sqlite3.connect(path)
        for kernel, counter, n, total, avg in db.execute(
                "select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                "group by kernel_name, counter_name"):
            short = kernel.split("(")[0].replace("void ", "")
            out.setdefault(short, {})[counter] = {"launches": n, "sum": total, "per_launch": avg}
        for kernel, n, avg in db.execute("select kernel_name, count(distinct dispatch_id), avg(end - start) "
                                         "from counters_collection group by kernel_name"):
            short = kernel.split("(")[0].replace("void ", "")
            out.setdefault(short, {})["avg_duration_ns_under_pmc"] = avg
    with open(out_json, "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
