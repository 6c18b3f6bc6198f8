#!/bin/bash
# Mid-size tuning of the symmetric pass on one GPU: shortest strip x K (guided divisor) x bodies per lane, sustained load.
# usage: tools/tune_mid.sh OUTFILE
out=${1:-gpurun_out/tune_mid.txt}
S="python tools/sweep.py --algos 2 --zeros 0"
{
for n in 32768 65536 131072; do
  for ms in 1 2 4; do for k in 4 6 8; do
    echo "N=$n MIN_SUB=$ms K=$k"; NBODY_SYM_K=$k NBODY_SYM_MIN_SUB=$ms $S --n $n --ipts 8,16 | tail -n +2
  done; done
done
} > $out 2>&1
