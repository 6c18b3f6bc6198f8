#!/bin/bash
# Same-box A/B of the lane-per-body walk: the next node's load issued before the term (default) against round 4's loop
# (libnbody_amd.lane_nopipe.so: make variant NAME=lane_nopipe EXTRA=-DNBODY_BH_LANE_NO_PIPELINE).  Frames of tools/bh_ticks.py.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for spec in "24576 200" "32768 200" "65536 200" "131072 100" "262144 100" "1048576 50"; do
  set -- $spec
  echo "next node fetched under the term: $(python3 tools/bh_ticks.py $1 $2 step 1.0 plummer)"
  echo "round 4's loop:                   $(NBODY_AMD_LIB=$ROOT/parallelnbody_amd/libnbody_amd.lane_nopipe.so python3 tools/bh_ticks.py $1 $2 step 1.0 plummer)"
done
