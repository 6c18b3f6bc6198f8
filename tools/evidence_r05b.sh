#!/bin/bash
# Round-5 evidence, second part (the even-share plan of the mid sizes), one call on the GPU box; copy what is to be judged from
# gpurun_out/r05b into profiles/:
#   PMC passes over the force kernel at N = 65536 (configs[1], equal masses, the library's default = even shares) and N = 32768;
#   PMC passes (HBM bytes) over the update kernel at N = 65536, distinct masses, guided strips against even shares;
#   the bench line (configs and bh rows) under rocprofv3 --kernel-trace --stats on this build.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r05b"
mkdir -p "$OUT"
cd "$ROOT"
bash tools/profile_kernel.sh "$OUT/pmc_force_n65536" 65536 f32 16 > "$OUT/pmc_force_n65536.log" 2>&1
bash tools/profile_kernel.sh "$OUT/pmc_force_n32768" 32768 f32 16 > "$OUT/pmc_force_n32768.log" 2>&1
rm -rf "$OUT"/pmc_force_n*/pmc[0-9]
echo "force kernel pmc done"
cd /tmp && export TMPDIR=/tmp
for EVEN in 0 1; do
  export NBODY_SYM_EVEN=$EVEN
  D="$OUT/pmc_update_n65536_even$EVEN"; mkdir -p "$D"
  i=0
  for set in "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$D/pmc$i" -o pmc -- python3 "$ROOT/tools/steps_plain.py" 65536 40 distinct > "$D/pmc${i}_stdout.txt" 2> "$D/pmc${i}_stderr.txt"
  done
  for k in update_sym_fused forces_sym_pk; do
    python3 "$ROOT/tools/pmc_kernel_means.py" "$D" $k "NBODY_SYM_EVEN=$EVEN tools/steps_plain.py 65536 40 distinct" > "$OUT/pmc_${k}_n65536_distinct_even$EVEN.txt"
  done
  rm -rf "$D"
done
unset NBODY_SYM_EVEN
echo "update kernel pmc done"
cd "$ROOT"
bash tools/profile_bench.sh "$OUT/bench" 16 > "$OUT/profile_bench.log" 2>&1
echo "bench profile done"
ls "$OUT"
