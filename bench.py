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


def parity_check(frames):
    """``frames`` = [(label, cube, gpu_result)]: the oracle's result for the same bytes, max relative error per frame."""
    from oracle import oracle_np as O
    errs = {}
    for label, cube, got in frames:
        ref = O.fft3d_windowed(cube, A)
        errs[label] = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    return errs


def cpu_baseline(seconds: float = 12.0):
    """Oracle (float64 NumPy restatement of the reference chain) timed on one host core."""
    from mmwave_radar_processing_amd import synth
    from oracle import oracle_np as O
    cubes = [synth.synth_cube(1000 + i) for i in range(4)]
    O.fft3d_windowed(cubes[0], A)     # warms numpy's FFT plan cache
    n, t0 = 0, time.perf_counter()
    while True:
        O.fft3d_windowed(cubes[n % len(cubes)], A)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and n >= 8:
            break
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames of the same synthetic 12x256x128 workload through oracle_np.fft3d_windowed "
                      f"(float64 NumPy, single thread) in {dt:.1f} s"}


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


def host_cores() -> int:
    """Physical cores this process may use (the pool has one single-threaded process per physical core)."""
    allowed = len(os.sched_getaffinity(0))
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or allowed
    except Exception:
        phys = allowed
    return max(1, min(allowed, phys))


def cpu_baseline_pool(per_proc_frames: int = 16):
    """The embarrassingly-parallel bound of the CPU path: one process per physical host core, one frame per task."""
    import multiprocessing as mp
    procs = host_cores()
    with mp.get_context("spawn").Pool(procs) as pool:
        pool.map(_cpu_worker, [(2000 + i, 1) for i in range(procs)])          # warm-up / imports
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(2000 + i, per_proc_frames) for i in range(procs)])
        dt = time.perf_counter() - t0
    n = procs * per_proc_frames
    return {"value": n / dt, "unit": "frames/s", "cores": procs, "kind": "port",
            "sample": f"{n} frames over a {procs}-process pool (one single-threaded NumPy process per physical host "
                      f"core; {len(os.sched_getaffinity(0))} logical CPUs allowed) in {dt:.1f} s"}


def baseline_metric() -> str:
    """BASELINE.json's metric string (the driver matches on it); falls back to the same text if the file is absent."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as fh:
            return json.load(fh)["metric"]
    except (OSError, KeyError, ValueError):
        return "radar frames/s on 256\u00d7128\u00d712 ADC cube, 1/2/4/8 GPU; % HBM roofline"


def chain_plan(ctx, n_frames: int) -> dict:
    """The schedule mmw_chain3d uses for this batch, asked of the library itself (mmw_diag_chain_plan)."""
    import ctypes
    from mmwave_radar_processing_amd import _lib
    plan = (ctypes.c_int * 8)()
    _lib.check(ctx.lib.mmw_diag_chain_plan(ctx.handle, n_frames, V, S, C, A, 0, plan))
    return {"overlapped": bool(plan[0]), "frames_per_launch_max": plan[1], "ring": plan[2], "rd_cus": plan[3],
            "rd_planes_per_frame": plan[4], "device_sync": bool(plan[6]), "ring_frames": plan[7]}


def insitu_ceiling(ctx, d_buf, nbytes: int) -> dict:
    """What plain streaming kernels reach on this device, right now: 16-B/lane grid-stride kernels over the whole (already
    checked) output buffer -- up to 20 GiB: a launch as long as the kernels it is compared with; over 4 GiB, as in rounds
    1-3, ramp-up and drain cost the write stream a tenth of its rate (5.1-5.2 against 5.8 TB/s) --, swept over the grid size,
    best of each kind reported: the practical ceiling next to the 8 TB/s spec peak.  mode 1 = plain stores, 3 = non-temporal
    stores, 6 = the angle kernel's store pattern without its arithmetic, 0 = copy, 2 = read."""
    from mmwave_radar_processing_amd import _lib
    nb = min(nbytes, 20 << 30) // (64 * 16384 * 16) * (64 * 16384 * 16)      # whole 16-MiB "frames" of the pattern kernel
    best = {}
    for name, mode in (("write", 1), ("write_nt", 3), ("write_angle_pattern", 6), ("copy", 0), ("read", 2)):
        span = nb // 2 if mode == 0 else nb
        for per_cu in (2, 4, 8, 16, 32):
            blocks = per_cu * 256
            call = lambda: _lib.check(ctx.lib.mmw_diag_membw(ctx.handle, d_buf.ptr, d_buf.ptr + (span if mode == 0 else 0),
                                                             span, mode, blocks))
            call()
            ctx.sync()
            ctx.timer_start()
            for _ in range(3):
                call()
            gbs = nb / (ctx.timer_stop() / 3) / 1e6
            if gbs > best.get(name + "_GBs", 0.0):
                best[name + "_GBs"] = gbs
                best[name + "_blocks_per_cu"] = per_cu
    return best


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


DET_CFAR = dict(kind=0, train=(4, 4), guard=(2, 2), pfa=1e-5)      # CaCFAR2D((4,4),(2,2),1e-5), SURVEY.md 8d
# the detector the reference's shipped configs run: OsCFAR2D((5,5),(3,2), rho 0.7, alpha 2) (gui_configs/processor_params.yaml:40-47)
OS_CFAR = dict(kind=1, train=(5, 5), guard=(3, 2), rho=0.7, alpha=2.0)
AZ_ANT, EL_ANT = list(range(8)), [8, 9, 10, 11]
DET_CAP = 2048
SUB_WARMUP = 25         # untimed calls in front of the timed steps of the `detect` / `detect_os` sub-records
DET_BYTES_PER_FRAME = 2 * CUBE_BYTES + S * C * 4    # RD cube in + out, antenna-0 magnitude (+ 8 B / detection)


def ca_alpha(n_train: int, pfa: float) -> float:
    return n_train * (pfa ** (-1.0 / n_train) - 1.0)       # detectors/base.py:293


class DetectWorkload:
    """BASELINE configs[2]: RD (all antennas) + float64 |RD| of antenna 0 + CA-CFAR + ordered compaction + exact
    azimuth / elevation argmax for every frame of the resident batch (FramePipeline.point_clouds without the host
    table look-ups)."""

    def __init__(self, ctx, F, path="fused", cfar=None):
        from mmwave_radar_processing_amd import _lib
        self.cfar = cfar = cfar or DET_CFAR
        if cfar["kind"] != 0:
            path = "float64"            # OS-CFAR: float64 CFAR plane + exact argmax (worst-case bound, dense float64 refinement)
        self.ctx, self.F, self._lib, self.path = ctx, F, _lib, path
        n = S * C
        self.d_rd = ctx.alloc(F * CUBE_BYTES)
        self.d_dets, self.d_cnt = ctx.alloc(F * DET_CAP * 8), ctx.alloc(F * 4)
        self.d_l1, self.d_az, self.d_el = ctx.alloc(F * V * 4), ctx.alloc(F * DET_CAP * 4), ctx.alloc(F * DET_CAP * 4)
        if path == "fused":
            self.d_mag32 = ctx.alloc(F * n * 4)
        else:
            self.d_mag, self.d_mask = ctx.alloc(F * n * 8), ctx.alloc(F * n)
        (tr, td), (gr, gd) = cfar["train"], cfar["guard"]
        n_train = (2 * (tr + gr) + 1) * (2 * (td + gd) + 1) - (2 * gr + 1) * (2 * gd + 1)
        if cfar["kind"] == 0:
            self.scale, self.k_rank = ca_alpha(n_train, cfar["pfa"]), 0
        else:
            self.scale, self.k_rank = cfar["alpha"], max(1, min(int(cfar["rho"] * n_train), n_train))     # detectors/os_cfar.py:131-132
        self.az, self.n_az = _lib.int_array(AZ_ANT)
        self.el, self.n_el = _lib.int_array(EL_ANT)

    def step(self, d_in, stats=None):
        L, h, lib, F = self.ctx.lib, self.ctx.handle, self._lib, self.F
        (tr, td), (gr, gd) = self.cfar["train"], self.cfar["guard"]
        kind = self.cfar["kind"]
        if self.path == "fused":
            # one call: RD + screened CFAR (undecided cells settled in float64) + ordered detections + az / el argmax
            lib.check(L.mmw_detect_points(h, d_in.ptr, self.d_rd.ptr, self.d_l1.ptr, self.d_mag32.ptr, self.d_dets.ptr,
                                          self.d_cnt.ptr, self.d_az.ptr, self.d_el.ptr, F, V, S, C, kind, tr, td, gr, gd,
                                          self.scale, 0, DET_CAP, self.az, self.n_az, 1, self.el, self.n_el, 0, A, stats))
            return
        lib.check(L.mmw_detect_batch(h, d_in.ptr, self.d_rd.ptr, self.d_mag.ptr, self.d_mask.ptr, self.d_dets.ptr,
                                     self.d_cnt.ptr, self.d_l1.ptr, F, V, S, C, kind, tr, td, gr, gd, self.scale,
                                     self.k_rank, DET_CAP))
        for ant, n_ant, d_idx, shift in ((self.az, self.n_az, self.d_az, 1), (self.el, self.n_el, self.d_el, 0)):
            lib.check(L.mmw_angle_argmax_exact(h, d_in.ptr, self.d_l1.ptr, self.d_rd.ptr, self.d_dets.ptr, self.d_cnt.ptr,
                                               d_idx.ptr, F, V, S, C, DET_CAP, ant, n_ant, A, shift, None))

    def parity(self, d_in, frames):
        """Detections (values and order) and argmax bins of the picked frames against the oracle: mismatch counts."""
        from mmwave_radar_processing_amd import synth
        from oracle import oracle_np as O
        sc = O.cfg_scalars(synth.SYNTH_CFG_256x128x12)
        counts = self.d_cnt.download((self.F,), np.int32)
        out = {}
        for f in frames:
            cube = d_in.download((V, S, C), np.complex64, f * CUBE_BYTES)
            if self.cfar["kind"] == 0:
                _, dets_ref, az_ref, el_ref = O.point_cloud(cube, sc, AZ_ANT, EL_ANT)
            else:
                dets_ref, az_ref, el_ref = os_point_indices(O, cube, self.cfar)
            n = int(counts[f])
            dets = self.d_dets.download((n, 2), np.int32, f * DET_CAP * 8).astype(np.int64)
            ok = n == dets_ref.shape[0] and np.array_equal(dets, dets_ref)
            bad_idx = -1
            if ok and n:
                az = self.d_az.download((n,), np.int32, f * DET_CAP * 4)
                el = self.d_el.download((n,), np.int32, f * DET_CAP * 4)
                bad_idx = int(np.count_nonzero(az != az_ref) + np.count_nonzero(el != el_ref))
            out[f"frame{f}"] = {"detections": n, "detection_indices_identical": bool(ok),
                                "argmax_index_differences": bad_idx if ok else None}
        return out, int(counts.sum())


def os_point_indices(O, cube, cfar):
    """Oracle: OS-CFAR 2-D detections (RangeDopplerDetector2D with os_cfar_2d) and the argmax bins of both antenna lists."""
    dets = O.rd_detect_2d_os(cube, cfar["train"], cfar["guard"], cfar["rho"], cfar["alpha"])
    if dets.shape[0] == 0:
        return dets, np.empty(0, int), np.empty(0, int)
    raw = O.range_doppler(cube)
    r, v = dets[:, 0].astype(int), dets[:, 1].astype(int)
    return dets, O.angle_argmax(raw, r, v, AZ_ANT, A, True)[0], O.angle_argmax(raw, r, v, EL_ANT, A, False)[0]


def cpu_baseline_detect(seconds: float = 12.0, cfar=None):
    from mmwave_radar_processing_amd import synth
    from oracle import oracle_np as O
    sc = O.cfg_scalars(synth.SYNTH_CFG_256x128x12)
    cubes = [synth.synth_cube(1000 + i) for i in range(4)]
    os_kind = bool(cfar and cfar["kind"] != 0)
    one = (lambda c: os_point_indices(O, c, cfar)) if os_kind else (lambda c: O.point_cloud(c, sc, AZ_ANT, EL_ANT))
    one(cubes[0])
    n, t0 = 0, time.perf_counter()
    while True:
        one(cubes[n % len(cubes)])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds and n >= (3 if os_kind else 8):
            break
    what = "OS-CFAR 2-D (sliding-window sort)" if os_kind else "CA-CFAR"
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} frames of the same synthetic 12x256x128 workload through the oracle (float64 NumPy RD + {what} + angle "
                      f"argmax, single thread) in {dt:.1f} s"}


def host_stream_record(ctx, chunk_frames: int = 128, n_chunks: int = 8):
    """PCIe-INCLUSIVE rate of the detection pipeline for frame loops whose cubes arrive on the host (never `value`):
    FramePipeline.stream over pinned host chunks, upload of chunk k + 1 on the copy queue while chunk k is processed;
    complex64 cubes and int16 I/Q raw cubes (half the bytes; layout not pinned by the reference)."""
    from mmwave_radar_processing_amd import synth
    from mmwave_radar_processing_amd.batch import FramePipeline
    from mmwave_radar_processing_amd.config_managers import ConfigManager
    from mmwave_radar_processing_amd.detectors import CaCFAR2D
    cm = ConfigManager()
    cm.load_cfg_text(synth.SYNTH_CFG_256x128x12)
    pipe = FramePipeline(cm, max_frames=chunk_frames, shape=(V, S, C), cfar=CaCFAR2D(DET_CFAR["train"], DET_CFAR["guard"], DET_CFAR["pfa"]),
                         az_antenna_idxs=AZ_ANT, el_antenna_idxs=EL_ANT, det_capacity=DET_CAP, ctx=ctx)
    pipe.synth(chunk_frames, seed0=4_000_000)
    cubes = pipe.cubes()
    nrx, ntx = 4, 3
    out = {"chunk_frames": chunk_frames, "chunks": n_chunks}
    work = lambda p: (p._alloc_detect(), p._detect_fused(True))[1]          # device results stay resident; counts come back
    for name, tx in (("c64", 0), ("i16", ntx)):
        if tx:
            raw = np.empty((chunk_frames, nrx, S, ntx * C), dtype=np.complex64)
            for t in range(ntx):
                raw[:, :, :, t::ntx] = cubes[:, t * nrx:(t + 1) * nrx]
            host = np.stack([raw.real, raw.imag], axis=-1).astype(np.int16)
        else:
            host = cubes
        pin = [ctx.host_array(host.shape, host.dtype) for _ in range(2)]
        for p in pin:
            p[...] = host
        ctx.sync()
        dets = 0
        for counts in pipe.stream([pin[i % 2] for i in range(2)], work=work, num_tx=tx, pinned=True):      # warm-up
            pass
        t0 = time.perf_counter()
        for counts in pipe.stream([pin[i % 2] for i in range(n_chunks)], work=work, num_tx=tx, pinned=True):
            dets += int(counts.sum())
        dt = time.perf_counter() - t0
        frames = n_chunks * chunk_frames
        out[name] = {"frames_per_s": frames / dt, "host_bytes_per_frame": host.nbytes // chunk_frames,
                     "h2d_GBs": frames * (host.nbytes / chunk_frames) / dt / 1e9, "detections": dets}
        for p in pin:
            ctx.host_free(p)
    pipe.bufs.free()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=1250, help="frames resident per GPU (10k-frame batch / 8 GPUs)")
    ap.add_argument("--workload", choices=("chain", "detect"), default="chain",
                    help="chain: range+Doppler+angle FFT chain (BASELINE configs[1], the headline); detect: RD + CA-CFAR + "
                         "point-cloud angle argmax (configs[2])")
    ap.add_argument("--detect-path", choices=("fused", "float64"), default="fused",
                    help="detect workload: mmw_detect_points (default) or round 2's mmw_detect_batch + mmw_angle_argmax_exact")
    ap.add_argument("--no-detect-record", action="store_true",
                    help="chain workload: skip the `detect` sub-record (BASELINE configs[2] timed after the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="record a HIP event pair around every n-th launch of each kernel inside the timed region; 0 = "
                         "auto: every launch when a step is one launch per stage, else every 7th (coprime with the "
                         "launches per step, so a short tail chunk is sampled in proportion)")
    _tj = [os.path.join(ROOT, "profiles", n) for n in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json")]
    ap.add_argument("--traffic-json", default=next((t for t in _tj if os.path.exists(t)), _tj[-1]))
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
    if world > ndev and not os.environ.get("MMW_BENCH_SHARE_DEVICES"):
        # two ranks on one device would also starve each other's device-synchronised chain
        raise SystemExit(f"bench.py: {world} ranks but only {ndev} visible device(s): one process per GPU is the contract "
                         "(set MMW_BENCH_SHARE_DEVICES=1 to rehearse on fewer devices; the figures are then meaningless)")
    ctx = _lib.Context(local_rank % ndev)
    info = _lib.device_info(ctx.device)
    F = args.frames
    detect = args.workload == "detect"
    d_in = ctx.alloc(F * CUBE_BYTES)
    d_out = None if detect else ctx.alloc(F * OUT_BYTES)
    work = DetectWorkload(ctx, F, args.detect_path) if detect else None
    # distinct frames per rank: seed0 offsets by the rank's first global frame index
    _lib.check(ctx.lib.mmw_synth_cubes(ctx.handle, d_in.ptr, F, V, S, C, 7_000_000 + rank * F, 8, 30.0))
    ctx.sync()
    plan = chain_plan(ctx, F)
    n_launch = 1 if detect else -(-F // plan["frames_per_launch_max"])    # kernel launches of each stage per step

    def step():
        if detect:
            work.step(d_in)
        else:
            _lib.check(ctx.lib.mmw_chain3d(ctx.handle, d_in.ptr, None, d_out.ptr, F, V, S, C, A, 0))

    def barrier():
        ctx.sync()
        if dist is not None:
            dist.barrier()

    def timed(step_fn, profile_every, warmup=None):
        """W untimed + exactly K timed steps between barriers; (max-over-ranks wall seconds, rank-local HIP-event ms)."""
        for _ in range(args.warmup if warmup is None else warmup):
            step_fn()
        barrier()
        if not args.no_profile:
            ctx.profile_reset()
            ctx.profile_enable(profile_every)
        t0 = time.perf_counter()
        ctx.timer_start()
        for _ in range(args.steps):
            step_fn()
        ev = ctx.timer_stop()
        ctx.sync()
        dt = time.perf_counter() - t0
        barrier()
        ctx.profile_enable(False)
        if dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, ev

    elapsed, ev_ms = timed(step, args.profile_every if args.profile_every > 0 else (1 if n_launch == 1 else 7))
    prof = {k: ctx.profile_get(k) for k in ("rd", "angle", "detect", "detect_tail", "detect_exact", "argmax_tail", "argmax_refine", "rd_help", "rd64", "cfar", "compact",
                                            "plane_l1", "argmax")} if not args.no_profile else {}
    # BASELINE configs[2] beside the headline: the detection pipeline on the same resident frames, same K / W
    det_extra = None
    if not detect and not args.no_detect_record:
        work = DetectWorkload(ctx, F, args.detect_path)
        # (sub-records: the detection pipeline's rate settles over its first ~20 calls -- 1.99, 1.79, 1.76, 1.74, 1.72 ms per call in
        #  blocks of six, profiles/r04_detect_probe.log -- so they warm up for SUB_WARMUP calls; the headline keeps --warmup)
        det_elapsed, det_ev = timed(lambda: work.step(d_in), 1, max(args.warmup, SUB_WARMUP))
        det_prof = {k: ctx.profile_get(k) for k in ("rd", "detect", "detect_tail", "detect_exact", "argmax_tail", "argmax_refine", "rd_help", "rd64", "cfar", "compact",
                                                    "plane_l1", "argmax")} if not args.no_profile else {}
        det_extra = (det_elapsed, det_ev, det_prof)
        work_os = DetectWorkload(ctx, F, cfar=OS_CFAR)
        os_elapsed, os_ev = timed(lambda: work_os.step(d_in), 1, max(args.warmup, SUB_WARMUP))
        os_prof = {k: ctx.profile_get(k) for k in ("rd", "rd64", "cfar", "compact", "plane_l1", "argmax")} if not args.no_profile else {}
        os_extra = (os_elapsed, os_ev, os_prof)

    def detect_fields(dt, ev, fam, wl=None):
        """frames/s, per-stage ms and the roofline of the range-Doppler stage of the detection pipeline."""
        v = world * F * args.steps / dt
        os_kind = wl is not None and wl.cfar["kind"] != 0
        rec = {"value": v, "unit": "frames/s", "ms_per_step": 1e3 * dt / args.steps, "hip_event_ms_per_step_rank0": ev / args.steps,
               "steps": args.steps, "warmup": args.warmup if detect else max(args.warmup, SUB_WARMUP),
               "algorithmic_bytes_per_frame": DET_BYTES_PER_FRAME,
               "hbm_frac_of_8TBs": v / world * DET_BYTES_PER_FRAME / (HBM_PEAK_GBS * 1e9),
               "path": wl.path if wl is not None else args.detect_path,
               "workload": ("range-Doppler of all antennas (float32) + OS-CFAR((5,5),(3,2), rho 0.7, alpha 2: the reference's GUI / "
                            "analysis default) on the float64 |RD| of antenna 0 + ordered detections + 8-antenna azimuth / 4-antenna "
                            "elevation argmax (float32, one lane per detection, worst-case error bound), flagged evaluations "
                            "refined from float64 cells (dense form: mmw_cells64.h)") if os_kind else
                           ("range-Doppler of all antennas (float32) + CA-CFAR((4,4),(2,2),1e-5) on antenna 0 (float32 screening "
                            "with the worst-case error band, undecided cells in float64) + ordered detections + 8-antenna azimuth / "
                            "4-antenna elevation argmax, float64-exact (BASELINE configs[2])")}
        if fam:
            rec["kernels_ms_per_step"] = {k: ms / max(n, 1) * (2 if k == "argmax" else 1) for k, (ms, n) in fam.items() if n}
            # consistency: the stages of the float64 / OS pipeline run back to back on one stream (sum of spans ~ step time); the
            # fused CA pipeline defers its tail (exact cells, insertion, refinement: spans `detect_exact`, `argmax_refine`) behind
            # the NEXT step's range-Doppler launch, so there the spans overlap and their sum exceeds the step time
            rec["kernels_ms_sum_over_ms_per_step"] = sum(rec["kernels_ms_per_step"].values()) / (1e3 * dt / args.steps)
            if not os_kind:
                rec["schedule"] = ("RD -> screening on the context stream; angle estimates (one lane per detection record), float64 "
                                   "refinement, exact cells + insertion on side queues, joined in front of the next call's screening "
                                   "(MMW_DETECT_DEFER_TAIL=1).  Behind a pending tail the range-Doppler planes are handed out by tickets "
                                   "to two launches of one kernel: num_cu - 40 workgroups at once (span `rd`), 40 more queued behind the "
                                   "tail (span `rd_help`, inside `rd`)")
            rd_ms, rd_n = fam.get("rd", (0.0, 0))
            if rd_n:
                avg_s = rd_ms * 1e-3 / rd_n
                achieved = F * 2 * CUBE_BYTES / avg_s / 1e9
                traffic, traffic_src = None, None
                if os.path.exists(args.traffic_json):
                    with open(args.traffic_json) as fh:
                        per_frame = (json.load(fh).get("detect") or {}).get("rd_bytes_per_frame")
                    if per_frame:
                        traffic = per_frame * F
                        traffic_src = ("static: PMC bytes per frame of this kernel from " + os.path.relpath(args.traffic_json, ROOT) +
                                       " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per the gfx950 "
                                       "note) x frames per launch; not measured in this run")
                rec["roofline"] = {"bound": "hbm", "kernel": "k_rd_fused_256x128_persist (range-Doppler of all 12 planes: the "
                                   "largest stage of the pipeline" + ("" if os_kind or "rd_help" not in fam or not fam["rd_help"][1] else
                                   "; two launches of it share the planes through a ticket counter -- avg_launch_us is the span of "
                                   "the first, num_cu - 40 workgroups, which the second, 40 workgroups behind the previous step's "
                                   "tail, lies within; a kernel trace lists both") + ")", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                                   "avg_launch_us": avg_s * 1e6,
                                   "launches": rd_n, "frames_per_launch": F,
                                   "algorithmic_bytes_per_launch": F * 2 * CUBE_BYTES}
        return rec

    if rank == 0:
        total_frames = world * F * args.steps
        value = total_frames / elapsed
        algo = DET_BYTES_PER_FRAME if detect else ALGO_BYTES_PER_FRAME
        out = {
            "metric": baseline_metric(),
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("IWR1843 synthetic 256x128x(4Rx x 3Tx) cube: range-Doppler of all antennas (float32) + "
                                    "CA-CFAR((4,4),(2,2),1e-5) on antenna 0 + ordered detections + float64-exact "
                                    "8-antenna azimuth / 4-antenna elevation argmax (BASELINE configs[2])") if detect else
                                   ("IWR1843 synthetic 256x128x(4Rx x 3Tx) cube: Hann range FFT + Doppler FFT + "
                                    "64-bin angle FFT -> complex64 [64,256,128] (BASELINE configs[1])"),
                       "frames_per_gpu": F, "cube": [V, S, C], "angle_bins": A,
                       "sharding": f"frame-sharded x{world}, no collective",
                       "schedule": (f"{args.detect_path}: stages back to back on one stream, whole batch per launch") if detect else plan,
                       "device": info["name"], "arch": info["arch"], "compute_units": info.get("num_cu")},
            "hip_event_ms_per_step_rank0": ev_ms / args.steps,
            "chain_hbm_frac_of_8TBs": value / world * algo / (HBM_PEAK_GBS * 1e9),
        }
        if detect:
            rec = detect_fields(elapsed, ev_ms, {k: v for k, v in prof.items() if k not in ("angle",)})
            for k in ("kernels_ms_per_step", "roofline"):
                if k in rec:
                    out[k] = rec[k]
        elif det_extra is not None:
            out["detect"] = detect_fields(*det_extra, wl=work)
            out["detect_os"] = detect_fields(*os_extra, wl=work_os)
        if not args.no_profile and not detect:
            ang_ms, ang_n = prof["angle"]
            rd_ms, rd_n = prof["rd"]
            if ang_n:
                # sampled launches: full chunks and the tail chunk are hit in proportion
                frames_per_launch = F / n_launch
                avg_s = ang_ms * 1e-3 / ang_n
                achieved = frames_per_launch * ALGO_BYTES_PER_FRAME / avg_s / 1e9
                traffic, traffic_src = None, None
                if os.path.exists(args.traffic_json):
                    with open(args.traffic_json) as fh:
                        per_frame = json.load(fh).get("angle_bytes_per_frame")
                    traffic = per_frame * frames_per_launch if per_frame else None
                    traffic_src = ("static: PMC bytes per frame from " + os.path.relpath(args.traffic_json, ROOT) +
                                   " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel, FETCH "
                                   "doubled per the gfx950 note) x frames per launch; not measured in this run")
                out["roofline"] = {"bound": "hbm",
                                   "kernel": ("k_angle64_sync" if plan["device_sync"] else "k_angle64") +
                                             " (angle FFT: reads the live RD planes, writes 64 angle planes)",
                                   "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                                   "avg_launch_us": avg_s * 1e6, "launches": ang_n,
                                   "frames_per_launch": frames_per_launch,
                                   "algorithmic_bytes_per_launch": frames_per_launch * ALGO_BYTES_PER_FRAME}
            if rd_n:
                fpl = F / n_launch
                avg_s = rd_ms * 1e-3 / rd_n
                rd_bytes = 2 * CUBE_BYTES * plan["rd_planes_per_frame"] / V      # planes read + planes written
                out["rd_kernel"] = {"avg_launch_us": avg_s * 1e6, "launches": rd_n,
                                    "achieved_GBs": fpl * rd_bytes / avg_s / 1e9,
                                    "algorithmic_bytes_per_launch": fpl * rd_bytes}
        # parity of this very run: first frame, a frame inside the last full launch and the batch's last frame (the
        # short tail launch), against the oracle on the SAME bytes -- not timed
        picks = sorted({0, max(0, (n_launch - 1) * plan["frames_per_launch_max"] - 1) if not detect else F // 2, F - 1})
        if detect:
            out["parity"], total_dets = work.parity(d_in, picks)
            out["detections_per_frame"] = total_dets / F
            parity_ok = all(v["detection_indices_identical"] and not v["argmax_index_differences"] for v in out["parity"].values())
        else:
            saved = [(f"frame{f}", d_in.download((V, S, C), np.complex64, f * CUBE_BYTES),
                      d_out.download((A, S, C), np.complex64, f * OUT_BYTES)) for f in picks]
            if "roofline" in out:
                out["roofline"]["insitu_streaming_ceiling"] = insitu_ceiling(ctx, d_out, F * OUT_BYTES)
            out["parity_max_rel_err"] = parity_check(saved)
            parity_ok = max(out["parity_max_rel_err"].values()) <= 1e-5
            if det_extra is not None:       # detections and argmax bins of three frames of the detect run against the oracle
                dpar, total_dets = work.parity(d_in, sorted({0, F // 2, F - 1}))
                out["detect"]["parity"] = dpar
                out["detect"]["detections_per_frame"] = total_dets / F
                parity_ok = parity_ok and all(v["detection_indices_identical"] and not v["argmax_index_differences"]
                                              for v in dpar.values())
                opar, os_dets = work_os.parity(d_in, sorted({0, F // 2, F - 1}))
                out["detect_os"]["parity"] = opar
                out["detect_os"]["detections_per_frame"] = os_dets / F
                parity_ok = parity_ok and all(v["detection_indices_identical"] and not v["argmax_index_differences"]
                                              for v in opar.values())
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_detect() if detect else cpu_baseline()
            if not detect:
                out["cpu_baseline_all_cores"] = cpu_baseline_pool()
                if det_extra is not None:
                    out["detect"]["cpu_baseline"] = cpu_baseline_detect(6.0)
                    out["detect_os"]["cpu_baseline"] = cpu_baseline_detect(6.0, OS_CFAR)
                    out["detect"]["host_stream_pcie_inclusive"] = host_stream_record(ctx)
        print(json.dumps(out))
        if not parity_ok:
            sys.exit("bench.py: GPU output differs from the oracle (spectra beyond 1e-5 / any index) -- the figure above is invalid")
    if dist is not None:
        with stdout_to_stderr():
            dist.barrier()
            dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
