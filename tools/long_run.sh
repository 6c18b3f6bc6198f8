#!/bin/bash
# BASELINE configs[4] in miniature on one GPU: N = 2^20 softened Plummer sphere, Kahan force accumulation, 1000 steps,
# energy logged every 100, once straight through and once with a checkpoint / resume in the middle; the two final states
# must be equal in every byte.   bash tools/long_run.sh OUTDIR [N [STEPS [PRECISION [ENERGY_EVERY]]]]
set -e
OUT="$1"; N="${2:-1048576}"; STEPS="${3:-1000}"; HALF=$((STEPS / 2)); PREC="${4:-f32_kahan}"; EVERY="${5:-100}"
mkdir -p "$OUT"
R="python -m parallelnbody_amd --plummer --n $N --eps 0.5 --dt 0.002 --precision $PREC --energy-every $EVERY --sync-energy"
echo "# $R --leapfrog-start --steps $STEPS --checkpoint straight.ckpt" > "$OUT/long_run.log"
$R --leapfrog-start --steps $STEPS --checkpoint "$OUT/straight.ckpt" >> "$OUT/long_run.log"
echo "# $R --leapfrog-start --steps $HALF --checkpoint half.ckpt" >> "$OUT/long_run.log"
$R --leapfrog-start --steps $HALF --checkpoint "$OUT/half.ckpt" >> "$OUT/long_run.log"
echo "# $R --resume half.ckpt --steps $HALF --checkpoint resumed.ckpt" >> "$OUT/long_run.log"
$R --resume "$OUT/half.ckpt" --steps $HALF --checkpoint "$OUT/resumed.ckpt" >> "$OUT/long_run.log"
if cmp "$OUT/straight.ckpt" "$OUT/resumed.ckpt"; then echo "# straight.ckpt and resumed.ckpt are identical ($(stat -c %s "$OUT/straight.ckpt") bytes, sha256 $(sha256sum "$OUT/straight.ckpt" | cut -c1-16))" >> "$OUT/long_run.log";
else echo "# MISMATCH between straight.ckpt and resumed.ckpt" >> "$OUT/long_run.log"; fi
python - "$OUT/long_run.log" >> "$OUT/long_run.log" <<'PY'
import json, sys
runs, cur = [], None
for ln in open(sys.argv[1]):
    if ln.startswith("#"):
        cur = []; runs.append(cur); continue
    try: r = json.loads(ln)
    except ValueError: continue
    if "total_synchronised" in r: cur.append((r["frame"], r["total_synchronised"]))
e0 = runs[0][0][1]
worst = max(abs(e - e0) / abs(e0) for _, e in runs[0])
print(f"# straight run: E(0) = {e0:.9e}; max |E(t) - E(0)| / |E(0)| over {len(runs[0])} samples = {worst:.3e}")
PY
rm -f "$OUT"/*.ckpt
tail -4 "$OUT/long_run.log"
