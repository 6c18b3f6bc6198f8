#!/bin/bash
# Same-box A/B of the lane-per-body walk: one XCD's workgroups take consecutive runs of the key order (default) against the
# plain numbering (libnbody_amd.noxcd.so: make variant NAME=noxcd EXTRA=-DNBODY_BH_NO_XCD_MAP).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for spec in "8192 200 box" "8192 200 plummer" "16384 200 plummer" "20480 200 plummer" "24576 200 plummer" "32768 200 plummer" "65536 200 plummer" "131072 100 plummer" "262144 100 plummer" "1048576 50 plummer" "65536 200 box" "1048576 50 box"; do
  set -- $spec
  echo "runs per XCD:     $(python3 tools/bh_ticks.py $1 $2 step 1.0 $3)"
  echo "plain numbering:  $(NBODY_AMD_LIB=$ROOT/parallelnbody_amd/libnbody_amd.noxcd.so python3 tools/bh_ticks.py $1 $2 step 1.0 $3)"
done
