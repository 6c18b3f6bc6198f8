#!/bin/bash
# Same-box A/B: every kernel of a theta > 0 frame asks for its first independent words together with the frame's verdict (default)
# against the verdict first and everything else behind it (libnbody_amd.before_hoist.so: the library built from the commit before).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for spec in "2000 400 box" "4096 200 box" "8192 200 box" "16384 200 plummer" "32768 200 plummer" "65536 200 plummer" "131072 100 plummer" "262144 100 plummer" "1048576 50 plummer" "65536 200 box"; do
  set -- $spec
  echo "first loads with the verdict: $(python3 tools/bh_ticks.py $1 $2 step 1.0 $3)"
  echo "behind the verdict:           $(NBODY_AMD_LIB=$ROOT/parallelnbody_amd/libnbody_amd.before_hoist.so python3 tools/bh_ticks.py $1 $2 step 1.0 $3)"
done
