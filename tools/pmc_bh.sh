#!/bin/bash
# PMC passes over the theta = 1 kernels of one system size (one counter set per run, never combined with tracing domains other
# than --kernel-trace), summarised per kernel by tools/pmc_bh_summary.py.
#   bash tools/pmc_bh.sh OUTDIR TAG N FRAMES [SCENE]      (on the GPU box; TAG names the summary: <TAG>_pmc_bh_kernels_n<N>_theta1.txt)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$1"; TAG="$2"; N="$3"; K="$4"; SCENE="${5:-}"
mkdir -p "$OUT"
OUT="$(cd "$OUT" && pwd)"
cd /tmp && export TMPDIR=/tmp
i=0
for cs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
          "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
          "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  rocprofv3 --pmc $cs --kernel-trace --output-format csv -d "$OUT/bh_pmc_n${N}_$i" -o pmc -- python3 "$ROOT/tools/bh_ticks.py" $N $K step 1.0 $SCENE > "$OUT/bh_pmc_n${N}_${i}_stdout.txt" 2>&1
  echo "pmc pass $i done"
done
python3 "$ROOT/tools/pmc_bh_summary.py" "$OUT" $N > "$OUT/${TAG}_pmc_bh_kernels_n${N}_theta1.txt" 2>&1
cat "$OUT/${TAG}_pmc_bh_kernels_n${N}_theta1.txt"
