#!/usr/bin/env python3
"""Sweep the chain's schedule knobs: one bench.py process per configuration (the library reads some knobs once per
process), compact one-line summaries on stdout, full JSON lines appended to --out.

    python tools/chain_sweep.py --out gpurun_out/sweep.jsonl "MMW_RD_CUS=128 MMW_CHAIN_RING=3" "MMW_RD_CUS=160" ...
Each positional argument is a space-separated list of NAME=VALUE pairs ("" = defaults).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=[""])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "chain_sweep.jsonl"))
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1250)
    args = ap.parse_args()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    for cfg in args.configs:
        env = dict(os.environ)
        for kv in cfg.split():
            k, v = kv.split("=", 1)
            env[k] = v
        t0 = time.time()
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup",
                            str(args.warmup), "--frames", str(args.frames), "--no-cpu-baseline"],
                           env=env, capture_output=True, text=True, timeout=600)
        line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else ""
        try:
            r = json.loads(line)
        except ValueError:
            print(f"[{cfg}] FAILED rc={p.returncode}: {p.stderr.strip()[-400:]}", flush=True)
            continue
        r["sweep_env"] = cfg
        with open(args.out, "a") as fh:
            fh.write(json.dumps(r) + "\n")
        roof = r.get("roofline", {})
        rd = r.get("rd_kernel", {})
        par = max(r.get("parity_max_rel_err", {"x": float("nan")}).values())
        print(f"[{cfg or 'defaults'}] {r['value'] / 1e3:7.1f} kf/s  chain {r['chain_hbm_frac_of_8TBs']:.3f}  "
              f"angle {roof.get('avg_launch_us', 0):7.1f} us ({roof.get('frac', 0):.3f})  rd {rd.get('avg_launch_us', 0):7.1f} us  "
              f"plan {r['config']['schedule']}  parity {par:.1e}  rc={p.returncode}  [{time.time() - t0:.0f} s]", flush=True)


if __name__ == "__main__":
    main()
