#!/bin/bash
# The coincident-body detector's table: how sparse should it be?  Whole steps and the update kernel's duration (rocprofv3
# --kernel-trace) by NBODY_SYM_DUP_FACTOR (slots = the power of two >= factor x N).   bash tools/ab_dup_factor.sh OUT N [N ...]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for N in "$@"; do
  for f in 2 4 8 16 32 64; do
    export NBODY_SYM_DUP_FACTOR=$f
    d="$OUT/t_n${N}_f$f"
    rocprofv3 --kernel-trace --output-format csv -d "$d" -o t -- python3 "$ROOT/tools/steps_plain.py" $N 600 distinct > "$d.stdout" 2> "$d.stderr"
    { echo "## N=$N table of >= $f x N slots: $(cat $d.stdout)"; python3 "$ROOT/tools/trace_gaps.py" "$d" 400 | grep -E "update_sym|block_pk|wall"; } >> "$OUT/ab_dup_factor.txt"
    rm -rf "$d"
  done
done
cat "$OUT/ab_dup_factor.txt"
