# Round-4 measurement collection (GPU box, repo root): everything under gpurun_out/r04/ ; tools/keep_r04.sh copies what is kept
# into profiles/.   bash tools/collect_r04.sh [part]   (part 1: tests + bench lines + traces, part 2: sweeps and micro-benchmarks)
set -o pipefail
mkdir -p gpurun_out/r04
part=${1:-1}
if [ "$part" = 1 ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r04/gpu_tests.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/r04/gpu_tests.log | tail -2
timeout -k 10 500 python bench.py > gpurun_out/r04/bench_line.json 2> gpurun_out/r04/bench.err; echo "bench rc=$?"
timeout -k 10 400 python bench.py --workload detect > gpurun_out/r04/bench_detect_line.json 2> gpurun_out/r04/bench_detect.err; echo "bench detect rc=$?"
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/r04/prof_bench -o bench -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-detect-record > gpurun_out/r04/bench_line_under_rocprof.json 2> gpurun_out/r04/prof_bench.err ); echo "rocprof bench rc=$?"
python tools/prof_summary.py stats $(find gpurun_out/r04/prof_bench -name "*.db" | head -1) gpurun_out/r04/bench_kernel_stats.csv && rm -rf gpurun_out/r04/prof_bench
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/r04/prof_detect -o detect -- python3 bench.py --workload detect --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r04/bench_detect_line_under_rocprof.json 2> gpurun_out/r04/prof_detect.err ); echo "rocprof detect rc=$?"
python tools/prof_summary.py stats $(find gpurun_out/r04/prof_detect -name "*.db" | head -1) gpurun_out/r04/detect_kernel_stats.csv && rm -rf gpurun_out/r04/prof_detect
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/r04/prof_os -o os -- python3 tools/os_pipeline.py --frames 256 --reps 3 > /dev/null 2> gpurun_out/r04/prof_os.err ); echo "rocprof os rc=$?"
python tools/prof_summary.py stats $(find gpurun_out/r04/prof_os -name "*.db" | head -1) gpurun_out/r04/os_kernel_stats.csv && rm -rf gpurun_out/r04/prof_os
timeout -k 10 300 python tools/os_pipeline.py > gpurun_out/r04/os_pipeline.json 2> gpurun_out/r04/os_pipeline.err; echo "os rc=$?"
( timeout -k 10 300 python tools/detect_cus.py --mag32 --splits 32,64,96 && timeout -k 10 100 python tools/detect_cus.py --clocks --splits 32 2>&1 | grep -A1 clocks ) > gpurun_out/r04/detect_schedules.log 2>&1; echo "schedules rc=$?"
python tools/bench_summary.py gpurun_out/r04/bench_line.json
else
# (the exactness sweeps: tools/collect_r04_sweeps.sh)
timeout -k 10 400 python tools/kbench.py --frames 1250 --reps 20 > gpurun_out/r04/kbench.json 2> gpurun_out/r04/kbench.err; echo "kbench rc=$?"
timeout -k 10 500 python tools/shapes_bench.py > gpurun_out/r04/shapes.json 2> gpurun_out/r04/shapes.err; echo "shapes rc=$?"
timeout -k 10 300 python tools/api_latency.py > gpurun_out/r04/api_latency.json 2> gpurun_out/r04/api_latency.err; echo "api rc=$?"
fi
