#!/bin/bash
# Whole-STEP rate (force pass + row folds + update: bench.py's `value`) at mid sizes against the guided divisor K and the
# shortest strip — the force-pass sweep alone does not see what the partial-sum segments cost update_sym_kernel.
out=${1:-gpurun_out/tune_step.txt}
{
for n in 32768 65536 131072 262144; do
  for ms in 1 2 4 8; do for k in 10 15 20 30 60; do
    r=$(NBODY_SYM_K_X10=$k NBODY_SYM_MIN_SUB=$ms python bench.py --bodies $n --steps 200 --warmup 5 --cpu-seconds 0 --no-distinct-row --settle-seconds 0.3 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms/step  %.2f %% of peak  force %.4f ms  update %.4f ms  items %d' % (r['ms_per_step'], r['value']*20/157.3e12*100, r['roofline']['avg_launch_ms'], r['roofline']['update_kernel_avg_ms'], r['config']['workgroups']))")
    echo "N=$n MIN_SUB=$ms K=$((k / 10)).$((k % 10))  $r"
  done; done
done
} > $out 2>&1
