#!/bin/bash
# Symmetric-kernel sweep by N and bodies per lane on one GPU (sustained load), fp32 / Kahan / fp64.
# usage: tools/sweep_sym.sh OUTFILE   (extra environment, e.g. NBODY_SYM_K=4, is passed through)
out=${1:-gpurun_out/sweep_sym.txt}
S="python tools/sweep.py --algos 2 --zeros 0"
{
echo "# fp32 exact, one-sided (tiled) kernel at the small sizes, for comparison"
for n in 8192 16384 32768; do python tools/sweep.py --algos 1 --zeros 0 --n $n --ipts 2,4 | tail -n +2; done
echo "# fp32 exact, symmetric, library defaults (bodies per lane, strip lengths) by N"
for n in 8192 16384 32768 65536 131072 262144 524288 1048576; do $S --n $n --ipts 0 | tail -n +2; done
echo "# fp32 exact, symmetric, by N and bodies per lane"
for n in 8192 16384 32768 65536 131072 262144 524288 1048576; do $S --n $n --ipts 2,4,8,16 | tail -n +2; done
echo "# Kahan"
for n in 65536 262144 1048576; do $S --n $n --precisions f32_kahan --ipts 2,4,8 | tail -n +2; done
echo "# Kahan softened eps=0.5 N=2^21"
$S --n 2097152 --precisions f32_kahan --ipts 8 --eps 0.5 | tail -n +2
echo "# fp64"
for n in 65536 262144 1048576; do $S --n $n --precisions f64 --ipts 2,4 | tail -n +2; done
} > $out 2>&1
