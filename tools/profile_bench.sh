#!/bin/bash
# Round evidence for profiles/: the bench line with its rocprofv3 kernel statistics, then PMC passes over the dominant
# kernel (tools/profile_kernel.sh: one counter set per run, never combined with tracing domains other than --kernel-trace).
#   bash tools/profile_bench.sh [outdir [bodies-per-lane]]       (on the GPU box; copy what you want judged from outdir into profiles/)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-$ROOT/gpurun_out/prof}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 \
  > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.txt"
tail -1 "$OUT/bench_line.json"
# PMC passes over the dominant kernel of that line
bash "$ROOT/tools/profile_kernel.sh" "$OUT" 1048576 f32 "${2:-16}"
