#!/bin/bash
# Sweep the CU split / chunk of the overlapped chain on one shape (SHAPE=V,S,C).
cd "$(dirname "$0")/.."
S=${SHAPE:-12,63,100}
run() { env "$@" python3 tools/chain_shape.py --shape $S --tag "[$*]"; }
run X=0
run MMW_CHAIN_PIPELINE=0
for cus in ${CUS:-0 64 80 96 112 128 144 160}; do
  for chunk in ${CHUNKS:-0 128 256}; do
    if [ $chunk = 0 ]; then run MMW_RD_CUS=$cus; else run MMW_RD_CUS=$cus MMW_CHAIN_CHUNK=$chunk; fi
  done
done
