#!/bin/bash
# A/B of the persistent (next-plane prefetch) and one-plane-per-workgroup variants of k_rd_mixed_ct.
cd "$(dirname "$0")/.."
for s in ${SHAPES:-12,63,70 12,63,100 12,64,40 12,70,40 12,90,80 12,100,30 12,254,50 4,127,32 12,90,100 12,100,100 12,120,126 12,130,50 12,200,40 12,63,127}; do
  for p in 0 1; do
    echo -n "persist=$p "; MMW_MIXED_CT_PERSIST=$p python3 tools/rd_prof.py --shape $s --frames ${FRAMES:-2048} --reps 5
  done
done
