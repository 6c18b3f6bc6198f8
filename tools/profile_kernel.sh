#!/bin/bash
# PMC passes over one force kernel (one counter set per run, never combined with tracing domains other than
# --kernel-trace), summarised by tools/pmc_summary.py.
#   bash tools/profile_kernel.sh OUTDIR N PRECISION BODIES_PER_LANE [EPS]      (on the GPU box)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$1"; N="$2"; PREC="$3"; BPL="$4"; EPS="${5:-0}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
KERNEL=forces_sym_pk_kernel; [ "$PREC" = f64 ] && KERNEL=forces_sym_f64_kernel
CMD="python3 $ROOT/tools/sweep.py --n $N --iters 1 --ipts $BPL --zeros 0 --algos 2 --precisions $PREC --eps $EPS"
i=0
for set in "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc$i" -o pmc -- $CMD > "$OUT/pmc${i}_stdout.txt" 2> "$OUT/pmc${i}_stderr.txt"
  echo "pmc pass $i done"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" $KERNEL "$N" "$BPL" "tools/sweep.py --n $N --iters 1 --ipts $BPL --zeros 0 --algos 2 --precisions $PREC --eps $EPS" > "$OUT/pmc_summary.txt"
cat "$OUT/pmc_summary.txt"
