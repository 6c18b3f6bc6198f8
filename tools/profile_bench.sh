#!/bin/bash
# Round evidence for profiles/: the bench line with its rocprofv3 kernel statistics, then PMC passes over the dominant
# kernel (one counter set per run, never combined with tracing domains other than --kernel-trace).
#   bash tools/profile_bench.sh [outdir [bodies-per-lane]]       (on the GPU box; copy what you want judged from outdir into profiles/)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/prof}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 \
  > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.txt"
tail -1 "$OUT/bench_line.json"
BPL="${2:-16}"   # bodies per lane of the profiled kernel
CMD="python3 $ROOT/tools/sweep.py --n 1048576 --iters 1 --ipts $BPL --zeros 0 --algos 2"
i=0
for set in "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" \
           "WRITE_SIZE"; do
  i=$((i + 1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/pmc$i" -o pmc -- $CMD > "$OUT/pmc${i}_stdout.txt" 2> "$OUT/pmc${i}_stderr.txt"
  echo "pmc pass $i done"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" forces_sym_pk_kernel "$BPL" > "$OUT/pmc_summary.txt"
cat "$OUT/pmc_summary.txt"
