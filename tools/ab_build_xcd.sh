#!/bin/bash
# Same-box A/B: bh_nodes_kernel and bh_sweep_chunks_kernel with one XCD's workgroups on consecutive bodies / chunks (default)
# against the plain numbering (libnbody_amd.build_noxcd.so: make variant NAME=build_noxcd EXTRA=-DNBODY_BH_BUILD_NO_XCD_RUNS).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for spec in "8192 200 box" "16384 200 plummer" "32768 200 plummer" "65536 200 plummer" "131072 100 plummer" "262144 100 plummer" "1048576 50 plummer" "65536 200 box"; do
  set -- $spec
  echo "runs per XCD:    $(python3 tools/bh_ticks.py $1 $2 step 1.0 $3)"
  echo "plain numbering: $(NBODY_AMD_LIB=$ROOT/parallelnbody_amd/libnbody_amd.build_noxcd.so python3 tools/bh_ticks.py $1 $2 step 1.0 $3)"
done
