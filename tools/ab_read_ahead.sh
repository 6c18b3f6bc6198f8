#!/bin/bash
# Same-box A/B of the LDS read-ahead in the fp32 symmetric kernel (NBODY_SYM_AHEAD = 0 nowhere / 1 equal-mass form /
# 2 everywhere registers allow): kernels_sym.hip is rebuilt on the GPU box for each setting and the N = 2^20 and
# N = 65536 force passes are timed under sustained load, equal-mass and general form (NBODY_SYM_NO_UNI=1).
#   bash tools/ab_read_ahead.sh [out]
out=${1:-gpurun_out/ab_read_ahead.txt}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
{
for v in 0 2 0 2; do
  rm -f parallelnbody_amd/csrc/kernels_sym.o
  make -C parallelnbody_amd/csrc EXTRA=-DNBODY_SYM_AHEAD=$v > /dev/null 2>&1
  for n in 1048576 65536; do
    echo "## NBODY_SYM_AHEAD=$v N=$n equal-mass form"
    python tools/sweep.py --n $n --iters 5 --ipts 16 --zeros 0 --algos 2 --precisions f32 | tail -1
    echo "## NBODY_SYM_AHEAD=$v N=$n general form"
    NBODY_SYM_NO_UNI=1 python tools/sweep.py --n $n --iters 5 --ipts 16 --zeros 0 --algos 2 --precisions f32 | tail -1
  done
  echo "## NBODY_SYM_AHEAD=$v N=2097152 Kahan eps 0.5 equal-mass form"
  python tools/sweep.py --n 2097152 --iters 2 --ipts 8 --zeros 0 --algos 2 --precisions f32_kahan --eps 0.5 | tail -1
done
rm -f parallelnbody_amd/csrc/kernels_sym.o
make -C parallelnbody_amd/csrc > /dev/null 2>&1
} > "$out" 2>&1
