#!/bin/bash
# Which walk for which size, on one box: frames of tools/bh_ticks.py with the lane-per-body walk forced from N on down
# (NBODY_BH_ROWS_MAX_N = largest system walked with windows; NBODY_BH_WAVE_MAX_N = largest walked with a wave per body).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for spec in "6144 200 box" "8192 200 box" "8192 200 plummer" "10240 200 plummer" "12288 200 plummer" "14336 200 plummer" "16384 200 plummer" "16384 200 box" "20480 200 plummer"; do
  set -- $spec
  echo "default:        $(python3 tools/bh_ticks.py $1 $2 step 1.0 $3)"
  echo "lane per body:  $(NBODY_BH_ROWS_MAX_N=0 python3 tools/bh_ticks.py $1 $2 step 1.0 $3)"
  echo "wave per body:  $(NBODY_BH_ROWS_MAX_N=1000000 NBODY_BH_WAVE_MAX_N=1000000 python3 tools/bh_ticks.py $1 $2 step 1.0 $3)"
done
