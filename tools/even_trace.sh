#!/bin/bash
# Kernel durations and gaps of whole steps (tools/steps_plain.py, distinct masses) under the guided and the even-share plan.
#   bash tools/even_trace.sh OUT N IPT [N IPT ...]     (on the GPU box)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  N=$1; IPT=$2; shift 2
  for EVEN in 0 1; do
    export NBODY_SYM_EVEN=$EVEN NBODY_SYM_IPT=$IPT
    d="$OUT/trace_n${N}_ipt${IPT}_even${EVEN}"
    rocprofv3 --kernel-trace --output-format csv -d "$d" -o t -- python3 "$ROOT/tools/steps_plain.py" $N 600 distinct > "$d.stdout" 2> "$d.stderr"
    { echo "## N=$N bodies per lane $IPT NBODY_SYM_EVEN=$EVEN: $(cat $d.stdout)"; python3 "$ROOT/tools/trace_gaps.py" "$d" 400; } >> "$OUT/even_trace.txt"
    rm -rf "$d"
  done
done
cat "$OUT/even_trace.txt"
