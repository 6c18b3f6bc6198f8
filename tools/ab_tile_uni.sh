#!/bin/bash
# Same-box A/B of the one-sided packed kernel with and without its equal-mass branch (kernels.hip rebuilt on the GPU box):
# force pass under sustained load, Plummer sphere (equal masses), exact / softened / Kahan, small to large N.
#   bash tools/ab_tile_uni.sh [out]
out=${1:-gpurun_out/ab_tile_uni.txt}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
{
for v in 0 1 0 1; do
  rm -f parallelnbody_amd/csrc/kernels.o
  make -C parallelnbody_amd/csrc EXTRA=-DNBODY_TILE_UNI=$v > /dev/null 2>&1
  for n in 8192 16384 65536 1048576; do
    for pe in "f32 0" "f32 0.5" "f32_kahan 0.5"; do
      set -- $pe
      echo "## NBODY_TILE_UNI=$v N=$n $1 eps $2"
      python tools/sweep.py --n $n --iters 3 --ipts 4 --zeros 0 --algos 1 --precisions $1 --eps $2 | tail -1
    done
  done
done
rm -f parallelnbody_amd/csrc/kernels.o
make -C parallelnbody_amd/csrc > /dev/null 2>&1
} > "$out" 2>&1
