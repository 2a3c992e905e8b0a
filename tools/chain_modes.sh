#!/bin/bash
# mmw_chain3d on one shape under the three schedules (serial / events / device-synchronised) and CU splits.
cd "$(dirname "$0")/.."
for S in ${SHAPES:-12,63,100 12,63,70 12,254,50 12,100,100 12,200,40 12,130,50 12,90,100 12,120,126 12,64,40 4,127,32}; do
  run() { env "$@" python3 tools/chain_shape.py --shape $S --frames ${FRAMES:-2048} --tag "[$*]"; }
  run MMW_CHAIN_PIPELINE=0
  run MMW_CHAIN_MODE=events
  for cus in ${CUS:-96 128 160}; do run MMW_CHAIN_MODE=sync MMW_RD_CUS=$cus; done
done
