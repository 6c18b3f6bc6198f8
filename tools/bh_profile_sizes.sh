#!/bin/bash
# theta = 1 frames by N: wall time per frame (nbody_step K frames in one call), then the same under rocprofv3 --kernel-trace --stats
# with a per-kernel table.   bash tools/bh_profile_sizes.sh OUTDIR TAG "N K" "N K" ...
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$(realpath -m "$1")"; TAG="$2"; shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec
  python3 "$ROOT/tools/bh_ticks.py" $1 $2 step | tee -a "$OUT/${TAG}_bh_frames_wall.txt"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_n$1" -o bh -- python3 "$ROOT/tools/bh_ticks.py" $1 $2 step > "$OUT/${TAG}_under_profiler_n$1.txt" 2>&1 || true
  f=$(find "$OUT/stats_n$1" -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then
    cp "$f" "$OUT/${TAG}_bh_kernel_stats_n$1_theta1.csv"
    { echo "== N=$1 (bh_ticks.py runs 3 + 3 x $2 frames)"; python3 "$ROOT/tools/bh_kernel_table.py" "$f" $((3 + 3 * $2)); } | tee -a "$OUT/${TAG}_bh_kernel_tables.txt"
  fi
done
