#!/bin/bash
# Same-box A/B: the block kernel's written-out DPP adds as ordered `asm volatile` behind one s_nop (the shipped build) against
# round 3's unordered form (make -C parallelnbody_amd/csrc variant NAME=dpp_unordered EXTRA=-DNBODY_BLOCK_DPP_UNORDERED).
# Whole steps without events, alternating, three rounds.
OUT=${1:-gpurun_out/ab_dpp_order.txt}
{
for r in 1 2 3; do
  for n in 2000 4096 8192 16384; do
    echo -n "ordered    "; python3 tools/steps_plain.py $n 2000
    echo -n "unordered  "; NBODY_AMD_LIB=$PWD/parallelnbody_amd/libnbody_amd.dpp_unordered.so python3 tools/steps_plain.py $n 2000
  done
done
} > $OUT 2>&1
