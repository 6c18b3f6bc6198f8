#!/bin/bash
# Whole steps without events by N: the library defaults (forces_block_pk_kernel up to N = 16384) against the paths it replaced.
OUT=${1:-gpurun_out/block_by_n.txt}
{
echo "# defaults"
for n in 500 1000 1500 2000 2560 3000 4096 5000 6000 7000 8192 8200 9216 10240 12288 14336 16384 17408 18432 20480; do python3 tools/steps_plain.py $n 1000; done
echo "# symmetric pass instead (NBODY_BLOCK_MAX_N=9216)"
for n in 9216 12288 14336 16384 17408 18432 20480; do NBODY_BLOCK_MAX_N=9216 python3 tools/steps_plain.py $n 1000; done
echo "# tile kernel + update instead (NBODY_BLOCK_MAX_N=1)"
for n in 2000 4096 6000 8192 10240 12288; do NBODY_BLOCK_MAX_N=1 python3 tools/steps_plain.py $n 1000; done
echo "# distinct masses, defaults"
for n in 2000 4096 8192 16384; do python3 tools/steps_plain.py $n 1000 distinct; done
echo "# softened, defaults"
for n in 2000 8192 16384; do python3 tools/steps_plain.py $n 1000 equal f32 0.01; done
} > $OUT 2>&1
