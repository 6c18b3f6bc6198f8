#!/bin/bash
# Round-3 evidence in one call on the GPU box (copy what you want judged from gpurun_out/r03 into profiles/):
#   the bench line under rocprofv3 with the PMC passes of its dominant kernel, the single-host line, the Barnes-Hut frames
#   (wall, kernel stats, PMC of the walks), whole steps by N with and without the per-kernel events.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r03"
mkdir -p "$OUT"
cd "$ROOT"
bash tools/profile_bench.sh "$OUT/bench" 16 > "$OUT/profile_bench.log" 2>&1
echo "bench profile done"
python3 bench.py --gpus 1 --host single --steps 5 --warmup 1 --cpu-seconds 0 > "$OUT/bench_line_single_host.json" 2> "$OUT/bench_single_host_stderr.txt"
echo "single host line done"
bash tools/profile_bh.sh "$OUT/bh" r03 > "$OUT/profile_bh.log" 2>&1
echo "bh profile done"
cd /tmp && export TMPDIR=/tmp
for spec in "2000 200" "1048576 10"; do
  set -- $spec
  i=0
  for cs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
            "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i + 1))
    rocprofv3 --pmc $cs --kernel-trace --output-format csv -d "$OUT/bh_pmc_n$1_$i" -o pmc -- python3 "$ROOT/tools/bh_ticks.py" $1 $2 step > "$OUT/bh_pmc_n$1_${i}_stdout.txt" 2>&1
  done
done
echo "bh pmc done"
cd "$ROOT"
bash tools/step_by_n.sh "$OUT/whole_step_by_n.txt"
{ echo "# tools/steps_plain.py: whole steps with the library defaults and NO per-kernel events (what a host runs), Plummer sphere"
  for n in 2000 4096 6000 8192 10240 12288 16384 20480 24576 32768 65536 131072; do python3 tools/steps_plain.py $n 1000; done
  echo "# the same bodies with distinct masses (general form of the kernels)"
  for n in 8192 16384 32768 65536; do python3 tools/steps_plain.py $n 1000 distinct; done; } > "$OUT/whole_step_no_events.txt" 2>&1
echo "steps done"
# last: where the small systems' build kernel spends its time — a tuning build with wall_clock64 stamps, built NEXT TO the shipped
# library (make variant: its own objects, its own .so), so that whatever runs from this checkout afterwards still tests the product
make -C parallelnbody_amd/csrc variant NAME=phase_clocks EXTRA=-DNBODY_BH_PHASE_CLOCKS > "$OUT/phase_build.log" 2>&1 || { echo "phase-clock build failed"; exit 1; }
{ NBODY_AMD_LIB=$ROOT/parallelnbody_amd/libnbody_amd.phase_clocks.so python3 tools/bh_phases.py 2000
  NBODY_AMD_LIB=$ROOT/parallelnbody_amd/libnbody_amd.phase_clocks.so python3 tools/bh_phases.py 4096; } > "$OUT/bh_build_phases.txt" 2>&1
echo "phases done"
ls "$OUT"
