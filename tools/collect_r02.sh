# Round-2 measurement collection (GPU box, repo root): writes everything under gpurun_out/r02/
set -o pipefail
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02/gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "schedule|sweep|bound divisor|passed|failed" gpurun_out/r02/gpu_tests.log | tail -12
MMW_SWEEP_FRAMES=2000 MMW_SWEEP_PROCS=32 timeout -k 10 600 python -m pytest tests/test_gpu_sweep.py -m gpu -x -q -s -k os_cfar > gpurun_out/r02/os_sweep.log 2>&1; grep -E "sweep|passed|failed" gpurun_out/r02/os_sweep.log
timeout -k 10 400 python bench.py > gpurun_out/r02/bench_line.json 2> gpurun_out/r02/bench.err; echo "bench rc=$?"
timeout -k 10 400 python bench.py --workload detect > gpurun_out/r02/bench_detect_line.json 2> gpurun_out/r02/bench_detect.err; echo "bench detect rc=$?"
timeout -k 10 400 python tools/kbench.py --frames 1250 --reps 20 > gpurun_out/r02/kbench.json 2> gpurun_out/r02/kbench.err
timeout -k 10 400 python tools/kbench.py --shape 12,63,100 --frames 2048 > gpurun_out/r02/kbench_63x100.json 2> gpurun_out/r02/kbench_63x100.err
timeout -k 10 500 python tools/shapes_bench.py > gpurun_out/r02/shapes.json 2> gpurun_out/r02/shapes.err
timeout -k 10 300 python tools/membw.py > gpurun_out/r02/membw.json 2> gpurun_out/r02/membw.err
timeout -k 10 300 python tools/api_latency.py > gpurun_out/r02/api_latency.json 2> gpurun_out/r02/api_latency.err
python - <<'PY'
import json
r=json.load(open("gpurun_out/r02/bench_line.json")); print("bench", r["value"], r["roofline"]["frac"], r["parity_max_rel_err"], r.get("cpu_baseline",{}).get("value"), r.get("cpu_baseline_all_cores",{}).get("value"))
r=json.load(open("gpurun_out/r02/bench_detect_line.json")); print("detect", r["value"], r["kernels_ms_per_step"], r["parity"], r.get("cpu_baseline",{}).get("value"))
print(open("gpurun_out/r02/api_latency.json").read())
PY

# ---- copy what is kept into profiles/ (tracked)
cp gpurun_out/r02/bench_line.json profiles/r02_bench_line.json
cp gpurun_out/r02/bench_detect_line.json profiles/r02_bench_detect_line.json
cp gpurun_out/r02/kbench.json profiles/r02_kbench.json
cp gpurun_out/r02/kbench_63x100.json profiles/r02_kbench_63x100.json
cp gpurun_out/r02/shapes.json profiles/r02_shapes.json
cp gpurun_out/r02/membw.json profiles/r02_membw.json
cp gpurun_out/r02/gpu_tests.log profiles/r02_gpu_tests.log
cp gpurun_out/r02/os_sweep.log profiles/r02_os_sweep.log
