#!/bin/bash
# forces_block_pk_kernel: rocprofv3 --kernel-trace --stats of whole steps and four PMC passes, at N = 8192 and 16384.
#   bash tools/profile_block.sh [outdir]     (on the GPU box; copy the summaries into profiles/)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$(realpath -m "${1:-$ROOT/gpurun_out/block}")"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for n in 2000 8192 16384; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_n$n" -o blk -- python3 "$ROOT/tools/steps_plain.py" $n 500 > "$OUT/steps_n${n}_under_profiler.txt" 2>&1
  f=$(find "$OUT/stats_n$n" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/r03_block_kernel_stats_n$n.csv"
  i=0; mkdir -p "$OUT/n$n"
  for cs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    rocprofv3 --pmc $cs --kernel-trace --output-format csv -d "$OUT/n$n/pmc$i" -o pmc -- python3 "$ROOT/tools/steps_plain.py" $n 100 > "$OUT/n$n/pmc${i}_stdout.txt" 2>&1
  done
  python3 "$ROOT/tools/pmc_kernel_means.py" "$OUT/n$n" forces_block_pk_kernel "tools/steps_plain.py $n 100" > "$OUT/r03_pmc_block_kernel_n$n.txt"
  cat "$OUT/r03_pmc_block_kernel_n$n.txt"
done
ls "$OUT"
