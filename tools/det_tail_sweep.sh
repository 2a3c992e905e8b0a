# deferred detection tail: CUs the next call's range-Doppler launch leaves free (MMW_DETECT_TAIL_CUS), each twice
for t in 0 32 40 48 0 32 64 ; do MMW_DETECT_TAIL_CUS=$t timeout -k 10 300 python bench.py --workload detect --no-cpu-baseline > gpurun_out/r04/det3_t$t.json 2> gpurun_out/r04/det3_t$t.err; echo "t=$t rc=$?"; python - <<PY
import json
d=json.load(open('gpurun_out/r04/det3_t$t.json'))
print($t, round(d['value']), round(d['ms_per_step'],4), round(d['chain_hbm_frac_of_8TBs'],4), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})
PY
done
MMW_DETECT_DEFER_TAIL=0 timeout -k 10 300 python bench.py --workload detect --no-cpu-baseline > gpurun_out/r04/det3_nodefer.json 2> gpurun_out/r04/det3_nodefer.err
python - <<PY
import json
d=json.load(open('gpurun_out/r04/det3_nodefer.json'))
print('nodefer', round(d['value']), round(d['ms_per_step'],4), round(d['chain_hbm_frac_of_8TBs'],4), {k:round(v,3) for k,v in d['kernels_ms_per_step'].items()})
PY
