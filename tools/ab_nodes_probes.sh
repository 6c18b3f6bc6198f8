#!/bin/bash
# Same-box A/B of bh_nodes_kernel's search for a cell's end: seven probes side by side and the stretch cut in eight (default)
# against steps of 1, 2, 4, ... and a halving, one load after the other
# (libnbody_amd.nodes_step.so: make variant NAME=nodes_step EXTRA=-DNBODY_BH_NODES_STEP_SEARCH).  Frames of tools/bh_ticks.py.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for spec in "8192 200 box" "16384 200 plummer" "32768 200 plummer" "65536 200 plummer" "131072 100 plummer" "262144 100 plummer" "1048576 50 plummer" "65536 200 box" "1048576 50 box"; do
  set -- $spec
  echo "probes side by side: $(python3 tools/bh_ticks.py $1 $2 step 1.0 $3)"
  echo "a step after another: $(NBODY_AMD_LIB=$ROOT/parallelnbody_amd/libnbody_amd.nodes_step.so python3 tools/bh_ticks.py $1 $2 step 1.0 $3)"
done
