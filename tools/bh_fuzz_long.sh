#!/bin/bash
# The Barnes-Hut fuzz tests run long: whole Ticks on random scenes (every byte of the records against the oracle after every
# call) and the force pass on random scenes (every bit), six seeds of NBODY_FUZZ_TRIALS scenes each.   tools/bh_fuzz_long.sh [OUT]
set -e -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/gpurun_out/bh_fuzz_long.txt}
TRIALS=${NBODY_FUZZ_TRIALS:-300}
cd "$ROOT"
echo "# NBODY_FUZZ_SEED=1..6 NBODY_FUZZ_TRIALS=$TRIALS python -m pytest tests/test_bh_gpu.py -m gpu -s -k fuzz" > "$OUT"
for s in 1 2 3 4 5 6; do
  NBODY_FUZZ_SEED=$s NBODY_FUZZ_TRIALS=$TRIALS timeout -k 10 900 python -m pytest tests/test_bh_gpu.py -x -q -s -k fuzz 2>&1 | tail -4 >> "$OUT"
  echo "seed $s done" >> "$OUT"
done
