# A/B of the symmetric kernel's j-side forms on one box (NBODY_SYM_JSCALAR=1: scalar running sums)
set -e
mkdir -p gpurun_out
S="python tools/sweep.py --iters 3 --zeros 0 --algos 2"
echo "== N = 2^20 fp32: packed travelling sums, ipt 4 and 8"; $S --n 1048576 --ipts 4,8 | tail -2
echo "== N = 2^20 fp32: scalar running sums, ipt 4"; NBODY_SYM_JSCALAR=1 $S --n 1048576 --ipts 4 | tail -1
echo "== N = 262144 Kahan: packed, ipt 2 and 4"; $S --n 262144 --precisions f32_kahan --ipts 2,4 | tail -2
echo "== N = 262144 Kahan: scalar, ipt 2 and 4"; NBODY_SYM_JSCALAR=1 $S --n 262144 --precisions f32_kahan --ipts 2,4 | tail -2
echo "== N = 65536 / 131072 fp32: ipt 2, 4, 8"; $S --n 65536 --ipts 2,4,8 | tail -3; $S --n 131072 --ipts 2,4,8 | tail -3
