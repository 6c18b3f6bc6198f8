#!/bin/bash
# Round-4 evidence in one call on the GPU box (copy what you want judged from gpurun_out/r04 into profiles/):
#   the bench line under rocprofv3 --kernel-trace --stats, the Barnes-Hut frames by N (wall, kernel statistics, per-kernel tables),
#   A/B lines of the round's switches on the same box, PMC passes over the theta = 1 kernels at N = 2000 and 2^20, and the build
#   kernel's phases from a tuning build that never replaces the shipped library (make variant NAME=phase_clocks; built beforehand).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r04"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -o bench -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 \
  > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.txt"
f=$(find "$OUT/bench_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/r04_bench_kernel_stats_symmetric.csv"
echo "bench done"
cd "$ROOT"
bash tools/bh_profile_sizes.sh "$OUT" r04 "2000 200" "4096 200" "8192 200" "65536 100" "262144 50" "1048576 30" > "$OUT/bh_profile.log" 2>&1
echo "bh profiles done"
{ echo "# theta = 1 frames, actor style (nbody_tick per frame: the step, the FParticle mirror, one host wait; from Python)"
  python3 tools/bh_ticks.py 2000 400 tick; python3 tools/bh_ticks.py 8192 200 tick; } > "$OUT/r04_bh_ticks_actor_style.txt" 2>&1
{ echo "# theta = 1 frames (tools/bh_ticks.py N K step), the round's switches on ONE box: default / NBODY_BH_SIZE_FROM_WALK=0 (a pass over the"
  echo "# positions per frame, round 3's bounds launch) / also NBODY_BH_WARM_SORT=0 (the cold sorts every frame: tiles + merge up to 131072, radix above)"
  for spec in "8192 400" "16384 200" "65536 200" "262144 100" "1048576 50"; do
    set -- $spec
    echo "default:            $(python3 tools/bh_ticks.py $1 $2 step)"
    echo "no Size from walk:  $(NBODY_BH_SIZE_FROM_WALK=0 python3 tools/bh_ticks.py $1 $2 step)"
    echo "cold sorts as well: $(NBODY_BH_SIZE_FROM_WALK=0 NBODY_BH_WARM_SORT=0 python3 tools/bh_ticks.py $1 $2 step)"
  done; } > "$OUT/r04_bh_warm_sort_ab.txt" 2>&1
echo "ab done"
cd /tmp
for spec in "2000 200" "1048576 10"; do
  set -- $spec
  i=0
  for cs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    rocprofv3 --pmc $cs --kernel-trace --output-format csv -d "$OUT/bh_pmc_n$1_$i" -o pmc -- python3 "$ROOT/tools/bh_ticks.py" $1 $2 step > "$OUT/bh_pmc_n$1_${i}_stdout.txt" 2>&1
  done
  python3 "$ROOT/tools/pmc_bh_summary.py" "$OUT" $1 > "$OUT/r04_pmc_bh_kernels_n$1_theta1.txt" 2>&1
done
echo "bh pmc done"
cd "$ROOT"
{ NBODY_AMD_LIB=$ROOT/parallelnbody_amd/libnbody_amd.phase_clocks.so python3 tools/bh_phases.py 2000
  NBODY_AMD_LIB=$ROOT/parallelnbody_amd/libnbody_amd.phase_clocks.so python3 tools/bh_phases.py 4096; } > "$OUT/r04_bh_build_phases.txt" 2>&1
echo "phases done"
ls "$OUT"
