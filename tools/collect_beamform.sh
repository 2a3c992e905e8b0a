# Beamformer measurements (GPU box, repo root): kbench entries + rocprofv3 kernel trace + --pmc pass of
# tools/beamform_prof.py, joined into gpurun_out/${R:-r04}/beamform.json (kept as profiles/${R:-r04}_beamform.json).
set -o pipefail
mkdir -p gpurun_out/${R:-r04}
K="bartlett_F1_256x256x64,bartlett_F16_256x256x64,bartlett_F16_256x256x900,capon_F1_12x512x128_T181,capon_F32_12x512x128_T181"
timeout -k 10 300 python tools/kbench.py --frames 64 --reps 20 --only $K > gpurun_out/${R:-r04}/kbench_beamform.json 2> gpurun_out/${R:-r04}/kbench_beamform.err; echo "kbench rc=$?"
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/${R:-r04}/prof_bf -o bf -- python3 tools/beamform_prof.py > /dev/null 2> gpurun_out/${R:-r04}/prof_bf.err ); echo "rocprof trace rc=$?"
python tools/prof_summary.py stats $(find gpurun_out/${R:-r04}/prof_bf -name "*.db" | head -1) gpurun_out/${R:-r04}/beamform_kernel_stats.csv && rm -rf gpurun_out/${R:-r04}/prof_bf
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/${R:-r04}/pmc_bf -o bf -- python3 tools/beamform_prof.py > /dev/null 2> gpurun_out/${R:-r04}/pmc_bf.err ); echo "rocprof pmc rc=$?"
python tools/beamform_summary.py gpurun_out/${R:-r04}/kbench_beamform.json $(find gpurun_out/${R:-r04}/pmc_bf -name "*.db" | head -1) gpurun_out/${R:-r04}/beamform.json profiles/r03_beamform.json > gpurun_out/${R:-r04}/beamform_summary.log && rm -rf gpurun_out/${R:-r04}/pmc_bf
python - <<'PY'
import json
import os; r = json.load(open("gpurun_out/" + os.environ.get("R", "r04") + "/beamform.json"))
for k, v in r["kbench"].items(): print(k, v)
for k, v in r["pmc"].items(): print(k, v)
print(r["mfma_valu_overlap_probe_TFLOPs_of_MFMA_work"])
PY
