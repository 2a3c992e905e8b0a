# copy the round-3 summaries that are kept into profiles/ (tracked)
set -e
for f in mixed_pmc pmc_traffic beamform bench_line bench_detect_line bench_detect_float64_line bench_line_under_rocprof bench_detect_line_under_rocprof kbench kbench_63x100 shapes api_latency; do
  [ -s gpurun_out/r03/$f.json ] && cp gpurun_out/r03/$f.json profiles/r03_$f.json
done
for f in bench_kernel_stats detect_kernel_stats beamform_kernel_stats; do [ -s gpurun_out/r03/$f.csv ] && cp gpurun_out/r03/$f.csv profiles/r03_$f.csv; done
for f in gpu_tests detect_phase_clocks sweep_10k os_sweep sequential_sweep argmax_os_sweep; do [ -s gpurun_out/r03/$f.log ] && cp gpurun_out/r03/$f.log profiles/r03_$f.log; done
ls -la profiles | grep r03
