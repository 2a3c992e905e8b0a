# Do float32 vector instructions execute UNDER float32 matrix instructions on this chip?  The counter the guide names for the
# question (MI355X_MICROARCH.md: SQ_VALU_MFMA_COEXEC_CYCLES) for the probe kernels of mmw_diag_mfma_peak (f32 MFMA alone, with 8 / 16
# v_fma_f32 after every MFMA, and the bf16-MFMA contrast case), for the two shipped planes whose 127-point level runs on f32 MFMA
# (k_rd_mixed_ct<254, 50>, <63, 127>) and for the Bartlett tile kernel.  On the GPU box, from the repo root:
#   bash tools/pmc_coexec.sh && python tools/pmc_coexec_summary.py gpurun_out/prof profiles/r04_coexec.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof
PMC="SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
for kind in 0 2 3 6 7; do
  timeout -k 10 120 rocprofv3 --pmc $PMC -d gpurun_out/prof/coexec_probe$kind -o p -- python3 tools/mfma_probe.py $kind > gpurun_out/prof/coexec_probe$kind.txt 2>&1 || exit 1
done
for shape in 12,254,50 12,63,127; do
  tag=$(echo $shape | tr ',' 'x')
  timeout -k 10 120 rocprofv3 --pmc $PMC -d gpurun_out/prof/coexec_rd_$tag -o p -- python3 tools/rd_prof.py --shape $shape --reps 2 > /dev/null 2>&1 || exit 1
done
timeout -k 10 180 rocprofv3 --pmc $PMC -d gpurun_out/prof/coexec_beamform -o p -- python3 tools/beamform_prof.py > /dev/null 2>&1 || exit 1
echo coexec passes done
