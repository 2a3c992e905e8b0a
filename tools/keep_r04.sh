# copy the round-4 summaries that are kept into profiles/ (tracked)
for f in bench_line bench_detect_line bench_line_under_rocprof bench_detect_line_under_rocprof os_pipeline kbench shapes api_latency; do
  [ -s gpurun_out/r04/$f.json ] && cp gpurun_out/r04/$f.json profiles/r04_$f.json
done
for f in bench_kernel_stats detect_kernel_stats os_kernel_stats; do [ -s gpurun_out/r04/$f.csv ] && cp gpurun_out/r04/$f.csv profiles/r04_$f.csv; done
for f in gpu_tests detect_schedules sweep_10k argmax_os_sweep; do [ -s gpurun_out/r04/$f.log ] && cp gpurun_out/r04/$f.log profiles/r04_$f.log; done
ls -la profiles | grep r04
