#!/usr/bin/env python3
"""Headline benchmark: radar frames/s of the range -> Doppler -> angle FFT chain on synthetic
256 x 128 x 12 ADC cubes (BASELINE.json configs[1], sharded per frame as in configs[4]).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU.  Every rank holds its own shard of `--frames` cubes resident in HBM (generated on
the device by mmw_synth_cubes before the timed region), a step is one pass of the chain over that shard.
Frames are independent, so there is NO data-path collective (weak scaling): torch.distributed (gloo) is
used only for the barriers around the timed region and the max-over-ranks of the elapsed time.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

V, S, C, A = 12, 256, 128, 64
CUBE_BYTES = V * S * C * 8                      # complex64 input cube
OUT_BYTES = A * S * C * 8                       # complex64 angle-range-Doppler cube
ALGO_BYTES_PER_FRAME = CUBE_BYTES + OUT_BYTES   # 19,922,944 B
HBM_PEAK_GBS = 8000.0                           # MI355X HBM3E spec (MI355X_MICROARCH.md)


def cpu_baseline(seconds: float = 12.0, gpu_frame=None):
    """Oracle (float64 NumPy restatement of the reference chain) timed on one host core.

    ``gpu_frame = (cube, gpu_result)``: frame 0 of the bench's own batch and what the GPU made of it; the oracle's
    result for that cube doubles as the parity check of the run (reported as ``parity_max_rel_err_frame0``)."""
    from mmwave_radar_processing_amd import synth
    from oracle import oracle_np as O
    cubes = [synth.synth_cube(1000 + i) for i in range(4)]
    ref0 = O.fft3d_windowed(cubes[0] if gpu_frame is None else gpu_frame[0], A)     # also warms numpy's FFT plan cache
    parity = None if gpu_frame is None else float(np.max(np.abs(gpu_frame[1] - ref0)) / np.max(np.abs(ref0)))
    n, t0 = 0, time.perf_counter()
    while True:
        O.fft3d_windowed(cubes[n % len(cubes)], A)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and n >= 8:
            break
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames of the same synthetic 12x256x128 workload through oracle_np.fft3d_windowed "
                      f"(float64 NumPy, single thread) in {dt:.1f} s",
            "parity_max_rel_err_frame0": parity}


def _cpu_worker(args):
    seed, n = args
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    from mmwave_radar_processing_amd import synth
    from oracle import oracle_np as O
    cube = synth.synth_cube(seed)
    t0 = time.perf_counter()
    for _ in range(n):
        O.fft3d_windowed(cube, A)
    return time.perf_counter() - t0


def cpu_baseline_pool(per_proc_frames: int = 24):
    """The embarrassingly-parallel bound of the CPU path: one process per host core, one frame per task."""
    import multiprocessing as mp
    procs = max(1, min(len(os.sched_getaffinity(0)), 64))
    with mp.get_context("spawn").Pool(procs) as pool:
        pool.map(_cpu_worker, [(2000 + i, 1) for i in range(procs)])          # warm-up / imports
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(2000 + i, per_proc_frames) for i in range(procs)])
        dt = time.perf_counter() - t0
    n = procs * per_proc_frames
    return {"value": n / dt, "unit": "frames/s", "cores": procs, "kind": "port",
            "sample": f"{n} frames over a {procs}-process pool (one single-threaded NumPy process per host core) "
                      f"in {dt:.1f} s"}


def baseline_metric() -> str:
    """BASELINE.json's metric string (the driver matches on it); falls back to the same text if the file is absent."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as fh:
            return json.load(fh)["metric"]
    except (OSError, KeyError, ValueError):
        return "radar frames/s on 256\u00d7128\u00d712 ADC cube, 1/2/4/8 GPU; % HBM roofline"


def chunk_frames(n_frames: int) -> int:
    """Frames per kernel launch of mmw_chain3d's default overlapped schedule (csrc/mmwgpu.hip)."""
    env = os.environ.get("MMW_CHAIN_CHUNK")
    if env:
        return max(1, min(int(env), n_frames))
    rd_cus = int(os.environ.get("MMW_RD_CUS", 256 * 5 // 8))
    auto = (250 << 20) // (2 * CUBE_BYTES)
    waves = auto * V // rd_cus
    if waves >= 1:
        auto = waves * rd_cus // V
    return max(1, min(auto, n_frames))


class stdout_to_stderr:
    """Route C-level stdout to stderr (gloo prints connection chatter there; stdout carries only the JSON line)."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1250, help="frames resident per GPU (10k-frame batch / 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--profile-every", type=int, default=7,
                    help="record a HIP event pair around every n-th launch of each kernel inside the timed region "
                         "(7 is coprime with the 32 launches of a 1250-frame step, so the short tail chunk is sampled "
                         "in proportion and the average matches rocprofv3's all-launch average)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist   # control plane only: barrier + max of the elapsed time
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with stdout_to_stderr():
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            dist.barrier()

    from mmwave_radar_processing_amd import _lib
    ndev = _lib.device_count()
    ctx = _lib.Context(local_rank % ndev)
    info = _lib.device_info(ctx.device)
    F = args.frames
    d_in = ctx.alloc(F * CUBE_BYTES)
    d_out = ctx.alloc(F * OUT_BYTES)
    # distinct frames per rank: seed0 offsets by the rank's first global frame index
    _lib.check(ctx.lib.mmw_synth_cubes(ctx.handle, d_in.ptr, F, V, S, C, 7_000_000 + rank * F, 8, 30.0))
    ctx.sync()

    def step():
        _lib.check(ctx.lib.mmw_chain3d(ctx.handle, d_in.ptr, None, d_out.ptr, F, V, S, C, A, 0))

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    if not args.no_profile:
        ctx.profile_reset()
        ctx.profile_enable(max(1, args.profile_every))
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        step()
    ev_ms = ctx.timer_stop()
    ctx.sync()
    elapsed = time.perf_counter() - t0
    barrier()
    ctx.profile_enable(False)

    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_frames = world * F * args.steps
        value = total_frames / elapsed
        out = {
            "metric": baseline_metric(),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "IWR1843 synthetic 256x128x(4Rx x 3Tx) cube: Hann range FFT + Doppler FFT + "
                                   "64-bin angle FFT -> complex64 [64,256,128] (BASELINE configs[1])",
                       "frames_per_gpu": F, "cube": [V, S, C], "angle_bins": A,
                       "sharding": f"frame-sharded x{world}, no collective",
                       "schedule": "overlapped: fused range-Doppler kernel on 5/8 of the CUs beside the angle kernel on "
                                   "3/8, 40-frame chunks, RD->angle intermediate resident in Infinity Cache "
                                   "(override: MMW_CHAIN_PIPELINE / MMW_CHAIN_CHUNK / MMW_RD_CUS)",
                       "device": info["name"], "arch": info["arch"]},
            "hip_event_ms_per_step_rank0": ev_ms / args.steps,
            "chain_hbm_frac_of_8TBs": value / world * ALGO_BYTES_PER_FRAME / (HBM_PEAK_GBS * 1e9),
        }
        if not args.no_profile:
            ang_ms, ang_n = ctx.profile_get("angle")
            rd_ms, rd_n = ctx.profile_get("rd")
            if ang_n:
                # sampled launches: full 40-frame chunks and the tail chunk are hit in proportion
                frames_per_launch = F / -(-F // chunk_frames(F))
                avg_s = ang_ms * 1e-3 / ang_n
                achieved = frames_per_launch * ALGO_BYTES_PER_FRAME / avg_s / 1e9
                traffic = None
                if os.path.exists(args.traffic_json):
                    with open(args.traffic_json) as fh:
                        per_frame = json.load(fh).get("angle_bytes_per_frame")
                    traffic = per_frame * frames_per_launch if per_frame else None
                out["roofline"] = {"bound": "hbm", "kernel": "k_angle64 (angle FFT, reads V planes / writes 64)",
                                   "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                                   "avg_launch_us": avg_s * 1e6, "launches": ang_n,
                                   "frames_per_launch": frames_per_launch,
                                   "algorithmic_bytes_per_launch": frames_per_launch * ALGO_BYTES_PER_FRAME}
            if rd_n:
                fpl = F / -(-F // chunk_frames(F))
                avg_s = rd_ms * 1e-3 / rd_n
                out["rd_kernel"] = {"avg_launch_us": avg_s * 1e6, "launches": rd_n,
                                    "achieved_GBs": fpl * 2 * CUBE_BYTES / avg_s / 1e9,
                                    "algorithmic_bytes_per_launch": fpl * 2 * CUBE_BYTES}
        # frame 0 of the bench's own data, saved for the parity check inside the cpu_baseline leg (not timed)
        cube0 = d_in.download((V, S, C), np.complex64)
        got0 = d_out.download((A, S, C), np.complex64)
        if "roofline" in out:
            # what plain streaming kernels reach on this device, right now (16-B/lane grid-stride write and copy over
            # 2 GiB of the output buffer, after frame 0 was saved): the practical ceiling next to the 8 TB/s spec peak
            nb = min(F * A * S * C * 8, 2 << 30) // 32 * 32
            insitu = {}
            for name, mode, moved in (("write", 1, nb), ("copy", 0, nb)):
                span = nb if mode == 1 else nb // 2
                call = lambda: _lib.check(ctx.lib.mmw_diag_membw(ctx.handle, d_out.ptr, d_out.ptr + (0 if mode == 1 else span),
                                                                 span, mode, 0))
                call()
                ctx.sync()
                ctx.timer_start()
                for _ in range(5):
                    call()
                insitu[name + "_GBs"] = moved / (ctx.timer_stop() / 5) / 1e6
            out["roofline"]["insitu_streaming_ceiling"] = insitu
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(gpu_frame=(cube0, got0))
            out["parity_max_rel_err_frame0"] = out["cpu_baseline"]["parity_max_rel_err_frame0"]
            out["cpu_baseline_all_cores"] = cpu_baseline_pool()
        print(json.dumps(out))
        if out.get("parity_max_rel_err_frame0") is not None and not out["parity_max_rel_err_frame0"] <= 1e-5:
            sys.exit("bench.py: GPU chain output differs from the oracle beyond 1e-5 -- the figure above is invalid")
    if dist is not None:
        with stdout_to_stderr():
            dist.barrier()
            dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
