#!/bin/bash
# What bh_sweep_top_kernel's time is made of (timing only: the experiment builds compute wrong sums):
#   make variant NAME=top1 EXTRA=-DNBODY_BH_TOP_EXPERIMENT=1   no fence + barrier between the levels
#   make variant NAME=top2 EXTRA=-DNBODY_BH_TOP_EXPERIMENT=2   no loads of the children's sums
#   make variant NAME=top3 EXTRA=-DNBODY_BH_TOP_EXPERIMENT=3   neither
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/top_exp"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "" top1 top2 top3; do
  for spec in "8192 200" "65536 100" "1048576 20"; do
    set -- $spec
    if [ -n "$lib" ]; then export NBODY_AMD_LIB="$ROOT/parallelnbody_amd/libnbody_amd.$lib.so"; else unset NBODY_AMD_LIB; fi
    rm -rf "$OUT/st"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/st" -o bh -- python3 "$ROOT/tools/bh_ticks.py" $1 $2 step > /dev/null 2>&1
    f=$(find "$OUT/st" -name "*kernel_stats.csv" | head -1)
    echo "${lib:-shipped} N=$1: $(python3 "$ROOT/tools/bh_kernel_table.py" "$f" $((3 + 3 * $2)) | grep "sweep_top")"
  done
done
rm -rf "$OUT/st"
