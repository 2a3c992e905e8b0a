#!/bin/bash
# The three shipped plane shapes with a 127-point level (254x50, 63x127, 127x32): mmw_chain3d at 2048 frames with the level on
# bfloat16 x 3 MFMAs (default) and on float32 MFMAs (MMW_BIGPRIME_BF16=0), for the cfgs' own antenna counts and for 12.
cd "$(dirname "$0")/.."
for S in ${SHAPES:-12,254,50 8,254,50 12,63,127 8,63,127 4,127,32 12,127,32}; do
  for bf in 1 0; do
    MMW_BIGPRIME_BF16=$bf python3 tools/chain_shape.py --shape $S --frames ${FRAMES:-2048} --tag "[MMW_BIGPRIME_BF16=$bf]"
    MMW_BIGPRIME_BF16=$bf python3 tools/chain_shape.py --rd --shape $S --frames ${FRAMES:-2048} --tag "[MMW_BIGPRIME_BF16=$bf]"
  done
done
