#!/bin/bash
# Kernel-level evidence for the theta > 0 path (the reference's shipped algorithm): rocprofv3 --kernel-trace --stats of K
# theta = 1 Ticks at N = 2000 (the shipped scene), 65536 and 2^20, plus wall time per frame with and without the profiler.
#   bash tools/profile_bh.sh [outdir] [tag]     (on the GPU box; copy the summaries into profiles/)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$(realpath -m "${1:-$ROOT/gpurun_out/bh}")"
TAG="${2:-r03}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for spec in "2000 200" "4096 200" "65536 50" "1048576 10"; do
  set -- $spec
  python3 "$ROOT/tools/bh_ticks.py" $1 $2 step | tee -a "$OUT/${TAG}_bh_wall.txt"
  python3 "$ROOT/tools/bh_ticks.py" $1 $2 tick | tee -a "$OUT/${TAG}_bh_wall.txt"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_n$1" -o bh -- python3 "$ROOT/tools/bh_ticks.py" $1 $2 step > "$OUT/${TAG}_bh_n$1_under_profiler.txt" 2>&1 || true
  f=$(find "$OUT/stats_n$1" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${TAG}_bh_kernel_stats_n$1_theta1.csv"
done
ls "$OUT"
