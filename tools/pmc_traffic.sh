# HBM-side traffic of the headline kernels (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes, program directly
# after --; FETCH doubled per the gfx950 note of MI355X_MICROARCH.md).  GPU box, repo root: bash tools/pmc_traffic.sh
#  * chain: `bench.py --steps 3 --warmup 0` under MMW_CHAIN_MODE=events MMW_ANGLE_QUEUES=1 (counter collection serialises
#    dispatches; the device-synchronised schedule needs its two launches to overlap) -- same loads and stores per frame.
#  * detection: `bench.py --workload detect --steps 3 --warmup 0` with MMW_DETECT_DEFER_TAIL=0 (every call joins its own tail: one
#    range-Doppler launch per call, the strided kernel -- the ticketed pair behind a pending tail moves the same bytes).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/${R:-r04}/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  MMW_CHAIN_MODE=events MMW_ANGLE_QUEUES=1 timeout -k 10 300 rocprofv3 --pmc $c -d gpurun_out/${R:-r04}/pmc/chain_$c -o p -- python3 bench.py --steps 3 --warmup 0 --no-cpu-baseline --no-profile --no-detect-record > /dev/null 2> gpurun_out/${R:-r04}/pmc/chain_$c.err || exit 1
  MMW_DETECT_DEFER_TAIL=0 timeout -k 10 300 rocprofv3 --pmc $c -d gpurun_out/${R:-r04}/pmc/det_$c -o p -- python3 bench.py --workload detect --steps 3 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> gpurun_out/${R:-r04}/pmc/det_$c.err || exit 1
done
python3 tools/pmc_traffic_summary.py gpurun_out/${R:-r04}/pmc gpurun_out/${R:-r04}/pmc_traffic.json && rm -rf gpurun_out/${R:-r04}/pmc
