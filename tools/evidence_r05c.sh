#!/bin/bash
# Round-5 evidence, third part (the theta > 0 frames after the hop word, the XCD runs and the first loads that go out with the
# verdict), one call on the GPU box; copy what is to be judged from gpurun_out/r05c into profiles/:
#   the bench line (configs, mid_sizes and bh rows) under rocprofv3 --kernel-trace --stats and the PMC passes over its dominant kernel;
#   the theta = 1 frames by N (wall, kernel statistics, per-kernel tables) and on the reference's kind of scene;
#   PMC passes over the theta = 1 kernels at N = 2000, 65536 and 2^20.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r05c"
mkdir -p "$OUT"
cd "$ROOT"
bash tools/profile_bench.sh "$OUT/bench" 16 > "$OUT/profile_bench.log" 2>&1
echo "bench profile done"
bash tools/bh_profile_sizes.sh "$OUT" r05c "2000 200" "4096 200" "8192 200" "65536 100" "262144 50" "1048576 30" > "$OUT/bh_profile.log" 2>&1
echo "bh profiles done"
{ echo "# theta = 1 frames on the reference's kind of scene at the large sizes (CreateSpacePoints(N, 1000): runaway bodies own Size within frames)"
  for spec in "65536 100" "262144 50" "1048576 30"; do set -- $spec; python3 tools/bh_ticks.py $1 $2 step 1.0 box; done
  echo "# ... and the mid sizes, Plummer spheres"
  for spec in "16384 200" "24576 200" "32768 200" "131072 100"; do set -- $spec; python3 tools/bh_ticks.py $1 $2 step 1.0 plummer; done
  echo "# actor style (nbody_tick per frame: the step, the FParticle mirror, one host wait; from Python)"
  python3 tools/bh_ticks.py 2000 400 tick; python3 tools/bh_ticks.py 8192 200 tick; } > "$OUT/r05c_bh_frames_wall_more.txt" 2>&1
echo "frames done"
for spec in "2000 200" "65536 60" "1048576 10"; do
  set -- $spec
  bash tools/pmc_bh.sh "$OUT/pmc_bh" r05c $1 $2 > "$OUT/pmc_bh_n$1.log" 2>&1
done
rm -rf "$OUT"/pmc_bh/bh_pmc_n*_[0-9] "$OUT"/stats_n* "$OUT"/bench/pmc[0-9]
echo "bh pmc done"
ls "$OUT"
