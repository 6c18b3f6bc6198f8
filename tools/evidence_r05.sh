#!/bin/bash
# Round-5 evidence in one call on the GPU box (copy what you want judged from gpurun_out/r05 into profiles/):
#   the bench line (with its `configs` and `bh` rows) under rocprofv3 --kernel-trace --stats and the PMC passes over its dominant
#   kernel; the theta = 1 frames by N and scene (wall, kernel statistics, per-kernel tables); PMC passes over the theta = 1 kernels at
#   N = 2000 and 2^20 (65536: tools/pmc_bh.sh, taken with the walk's A/B); one rank's share of a sharded theta = 1 frame.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r05"
mkdir -p "$OUT"
cd "$ROOT"
bash tools/profile_bench.sh "$OUT/bench" 16 > "$OUT/profile_bench.log" 2>&1
echo "bench profile done"
bash tools/bh_profile_sizes.sh "$OUT" r05 "2000 200" "4096 200" "8192 200" "65536 100" "262144 50" "1048576 30" > "$OUT/bh_profile.log" 2>&1
echo "bh profiles done"
{ echo "# theta = 1 frames on the reference's kind of scene at every size (CreateSpacePoints(N, 1000): runaway bodies own Size within frames)"
  for spec in "65536 100" "262144 50" "1048576 30"; do set -- $spec; python3 tools/bh_ticks.py $1 $2 step 1.0 box; done; } > "$OUT/r05_bh_frames_wall_box_scene.txt" 2>&1
{ echo "# theta = 1 frames, actor style (nbody_tick per frame: the step, the FParticle mirror, one host wait; from Python)"
  python3 tools/bh_ticks.py 2000 400 tick; python3 tools/bh_ticks.py 8192 200 tick; } > "$OUT/r05_bh_ticks_actor_style.txt" 2>&1
echo "frames done"
for spec in "2000 200" "1048576 10"; do
  set -- $spec
  bash tools/pmc_bh.sh "$OUT/pmc_bh" r05 $1 $2 > "$OUT/pmc_bh_n$1.log" 2>&1
done
echo "bh pmc done"
{ python3 tools/bh_shard_time.py 65536 100; python3 tools/bh_shard_time.py 1048576 30; } > "$OUT/r05_bh_rank_share_theta1.txt" 2>&1
echo "shares done"
ls "$OUT"
