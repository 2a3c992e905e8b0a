cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof
for variant in new direct; do
  if [ $variant = direct ]; then export MMWGPU_LIB=$GRAFT_REPO_ROOT/gpurun_ab/libmmwgpu_direct.so; else unset MMWGPU_LIB; fi
  python3 tools/rd_prof.py --shape 12,63,100
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d gpurun_out/prof/rdmix_${variant}_a -o p -- python3 tools/rd_prof.py --shape 12,63,100 --reps 2 > /dev/null 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/prof/rdmix_${variant}_b -o p -- python3 tools/rd_prof.py --shape 12,63,100 --reps 2 > /dev/null 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC -d gpurun_out/prof/rdmix_${variant}_c -o p -- python3 tools/rd_prof.py --shape 12,63,100 --reps 2 > /dev/null 2>&1 || echo "pass c failed"
done
