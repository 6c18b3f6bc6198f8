#!/bin/bash
# Bodies per wave of the lane-per-body walk (bh_walk_lane_kernel): 64 / 32 / 16 (NBODY_BH_LANE_SHIFT = 0 / 1 / 2), frames of
# tools/bh_ticks.py on ONE box; "auto" is what bh_lane_shift picked.  The record of a round-5 experiment that did NOT pay
# (profiles/r05_ab_lane_walk_bodies_per_wave.txt: never faster, up to 2.4x slower at 2^20): a wave-step costs the same whatever
# number of its lanes is at work, so fewer bodies to a wave only multiplies the instructions.  The switch is gone from the library.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for spec in "24576 200" "32768 200" "49152 200" "65536 200" "98304 100" "131072 100" "196608 100" "262144 100" "524288 50" "1048576 50"; do
  set -- $spec
  for sh in 0 1 2 auto; do
    if [ $sh = auto ]; then echo "auto:    $(python3 tools/bh_ticks.py $1 $2 step 1.0 plummer)"
    else echo "shift $sh: $(NBODY_BH_LANE_SHIFT=$sh python3 tools/bh_ticks.py $1 $2 step 1.0 plummer)"; fi
  done
done
