# Round-3 measurement collection (GPU box, repo root): writes everything under gpurun_out/r03/ ; tools/keep_r03.sh copies the
# summaries that are kept into profiles/.
set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r03/gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/r03/gpu_tests.log | tail -2
timeout -k 10 400 python bench.py > gpurun_out/r03/bench_line.json 2> gpurun_out/r03/bench.err; echo "bench rc=$?"
timeout -k 10 400 python bench.py --workload detect > gpurun_out/r03/bench_detect_line.json 2> gpurun_out/r03/bench_detect.err; echo "bench detect rc=$?"
timeout -k 10 400 python bench.py --workload detect --detect-path float64 --no-cpu-baseline > gpurun_out/r03/bench_detect_float64_line.json 2> gpurun_out/r03/bench_detect_f64.err; echo "bench detect f64 rc=$?"
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/r03/prof_bench -o bench -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03/bench_line_under_rocprof.json 2> gpurun_out/r03/prof_bench.err ); echo "rocprof bench rc=$?"
python tools/prof_summary.py stats $(find gpurun_out/r03/prof_bench -name "*.db" | head -1) gpurun_out/r03/bench_kernel_stats.csv && rm -rf gpurun_out/r03/prof_bench
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/r03/prof_detect -o detect -- python3 bench.py --workload detect --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r03/bench_detect_line_under_rocprof.json 2> gpurun_out/r03/prof_detect.err ); echo "rocprof detect rc=$?"
python tools/prof_summary.py stats $(find gpurun_out/r03/prof_detect -name "*.db" | head -1) gpurun_out/r03/detect_kernel_stats.csv && rm -rf gpurun_out/r03/prof_detect
timeout -k 10 400 python tools/kbench.py --frames 1250 --reps 20 > gpurun_out/r03/kbench.json 2> gpurun_out/r03/kbench.err; echo "kbench rc=$?"
timeout -k 10 400 python tools/kbench.py --shape 12,63,100 --frames 2048 > gpurun_out/r03/kbench_63x100.json 2> gpurun_out/r03/kbench_63x100.err
timeout -k 10 500 python tools/shapes_bench.py > gpurun_out/r03/shapes.json 2> gpurun_out/r03/shapes.err; echo "shapes rc=$?"
timeout -k 10 300 python tools/api_latency.py > gpurun_out/r03/api_latency.json 2> gpurun_out/r03/api_latency.err
MMW_PHASE_CLOCKS=1 timeout -k 10 200 python bench.py --workload detect --no-cpu-baseline --steps 3 --warmup 1 2>&1 >/dev/null | grep clocks | tail -2 > gpurun_out/r03/detect_phase_clocks.log
python - <<'PY'
import json
r=json.load(open("gpurun_out/r03/bench_line.json")); d=r["detect"]
print("bench", r["value"], r["roofline"]["frac"], r["parity_max_rel_err"], r.get("cpu_baseline",{}).get("value"), r.get("cpu_baseline_all_cores",{}).get("value"))
print("detect sub-record", d["value"], d["hbm_frac_of_8TBs"], d["kernels_ms_per_step"], d["parity"], d.get("host_stream_pcie_inclusive"))
r=json.load(open("gpurun_out/r03/bench_detect_line.json")); print("detect", r["value"], r["kernels_ms_per_step"], r["parity"], r.get("cpu_baseline",{}).get("value"))
r=json.load(open("gpurun_out/r03/bench_detect_float64_line.json")); print("detect float64 path", r["value"], r["kernels_ms_per_step"])
print(open("gpurun_out/r03/api_latency.json").read())
print(open("gpurun_out/r03/detect_phase_clocks.log").read())
PY
