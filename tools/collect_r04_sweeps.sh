# Round-4 exactness sweeps (GPU box, repo root) -> gpurun_out/r04/
mkdir -p gpurun_out/r04
MMW_SWEEP_FRAMES=10000 timeout -k 10 600 python -m pytest tests/test_gpu_sweep.py -x -q -s -k test_detection_indices > gpurun_out/r04/sweep_10k.log 2>&1; echo "sweep rc=$?"; tail -3 gpurun_out/r04/sweep_10k.log
MMW_SWEEP_FRAMES=640 timeout -k 10 400 python -m pytest tests/test_gpu_sweep.py -x -q -s -k "standalone_exact_argmax or os_cfar" > gpurun_out/r04/argmax_os_sweep.log 2>&1; echo "argmax / os sweep rc=$?"; tail -3 gpurun_out/r04/argmax_os_sweep.log
