#!/bin/bash
# What the fused update of the symmetric pass spends where: builds of update_sym_fused_kernel that return at once (1), after the
# row folds (2), or leave out the coincident-body detector's entry (3), next to the shipped library (`make variant`), timed by
# rocprofv3 --kernel-trace over whole steps.   bash tools/ab_update_parts.sh OUT N [N ...]      (on the GPU box; results of the
# variants' steps are NOT physics — only the kernel's duration is looked at)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/$1"; shift
mkdir -p "$OUT"
for v in 1 2 3; do make -s -C "$ROOT/parallelnbody_amd/csrc" variant NAME=upd$v EXTRA=-DNBODY_UPD_EXPERIMENT=$v -j8 > "$OUT/build_upd$v.log" 2>&1; done
cd /tmp && export TMPDIR=/tmp
for N in "$@"; do
  for v in 0 1 2 3; do
    if [ $v = 0 ]; then unset NBODY_AMD_LIB; else export NBODY_AMD_LIB="$ROOT/parallelnbody_amd/libnbody_amd.upd$v.so"; fi
    d="$OUT/t_n${N}_v$v"
    rocprofv3 --kernel-trace --output-format csv -d "$d" -o t -- python3 "$ROOT/tools/steps_plain.py" $N 600 distinct > "$d.stdout" 2> "$d.stderr"
    { echo "## N=$N update variant $v (0 = shipped; 1 returns at once; 2 after the folds; 3 without the detector's entry): $(cat $d.stdout)"; python3 "$ROOT/tools/trace_gaps.py" "$d" 400 | grep -E "update_sym|forces_sym|wall"; } >> "$OUT/ab_update_parts.txt"
    rm -rf "$d"
  done
done
cat "$OUT/ab_update_parts.txt"
