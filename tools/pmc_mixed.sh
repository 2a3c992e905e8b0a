# rocprofv3 --pmc passes for the compile-time mixed-radix RD kernels (63 x 100 and 254 x 50 planes): HBM-side traffic and
# LDS bank conflicts.  Run on the GPU box from the repo root: [SHAPES="12,254,50 12,63,127"] bash tools/pmc_mixed.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof
for shape in ${SHAPES:-12,63,100 12,254,50}; do
  tag=$(echo $shape | tr ',' 'x')
  timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/mix_${tag}_fetch -o p -- python3 tools/rd_prof.py --shape $shape --reps 2 > /dev/null 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/mix_${tag}_write -o p -- python3 tools/rd_prof.py --shape $shape --reps 2 > /dev/null 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY -d gpurun_out/prof/mix_${tag}_sq -o p -- python3 tools/rd_prof.py --shape $shape --reps 2 > /dev/null 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d gpurun_out/prof/mix_${tag}_mfma -o p -- python3 tools/rd_prof.py --shape $shape --reps 2 > /dev/null 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --pmc GRBM_GUI_ACTIVE -d gpurun_out/prof/mix_${tag}_grbm -o p -- python3 tools/rd_prof.py --shape $shape --reps 2 > /dev/null 2>&1 || exit 1
done
