#!/bin/bash
# Same-box A/B of the lane-per-body walk: the node's hop word (where to go, the level's threshold ready-made: default) against the
# packed node word unpacked inside the step with the threshold read from LDS
# (libnbody_amd.lane_meta.so: make variant NAME=lane_meta EXTRA=-DNBODY_BH_LANE_META_WORD).  Frames of tools/bh_ticks.py.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for spec in "24576 200" "32768 200" "65536 200" "131072 100" "262144 100" "1048576 50"; do
  set -- $spec
  echo "hop word:          $(python3 tools/bh_ticks.py $1 $2 step 1.0 plummer)"
  echo "packed word + LDS: $(NBODY_AMD_LIB=$ROOT/parallelnbody_amd/libnbody_amd.lane_meta.so python3 tools/bh_ticks.py $1 $2 step 1.0 plummer)"
done
echo "box scene (CreateSpacePoints):"
for spec in "65536 200" "1048576 50"; do
  set -- $spec
  echo "hop word:          $(python3 tools/bh_ticks.py $1 $2 step 1.0 box)"
  echo "packed word + LDS: $(NBODY_AMD_LIB=$ROOT/parallelnbody_amd/libnbody_amd.lane_meta.so python3 tools/bh_ticks.py $1 $2 step 1.0 box)"
done
