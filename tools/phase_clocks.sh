#!/bin/bash
# Phase clocks of k_rd_mixed_ct (workgroup 0) for the shapes given, plus the undisturbed timing.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for s in ${SHAPES:-12,254,50 12,63,127 4,127,32 12,100,100 12,200,40 12,63,100}; do
  MMW_PHASE_CLOCKS=1 python3 tools/rd_prof.py --shape $s --frames 512 --reps 1 2>&1 | sort | uniq -c
  python3 tools/rd_prof.py --shape $s --frames 2048 --reps 5
done
