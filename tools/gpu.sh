#!/bin/bash
# tools/gpu.sh <tag> <timeout_s> '<command>' : run a command on the GPU box through gpurun, retrying while no slot is free
# (exit 3 = nothing charged); output of the call in gpurun_out/r04/<tag>.call.txt
tag=$1; to=$2; shift 2
mkdir -p gpurun_out/r04
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /usr/local/graft/bin/gpurun --timeout "$to" -- "mkdir -p gpurun_out/r04 && $*" > gpurun_out/r04/$tag.call.txt 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "rc=$rc" >> gpurun_out/r04/$tag.call.txt; exit $rc; fi
  sleep 90
done
echo "rc=3 (gave up)" >> gpurun_out/r04/$tag.call.txt
