#!/bin/bash
# Same-box A/B of the row folds' unroll (segment loads in flight per thread): kernels_sym*.hip rebuilt on the GPU box per
# setting, whole steps at mid sizes through bench.py.     bash tools/ab_fold_unroll.sh [out]
out=${1:-gpurun_out/ab_fold_unroll.txt}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
{
for u in 4 8 16 4 8 16; do
  rm -f parallelnbody_amd/csrc/kernels_sym.o parallelnbody_amd/csrc/kernels_sym64.o
  make -C parallelnbody_amd/csrc EXTRA=-DNBODY_SYM_FOLD_UNROLL=$u > /dev/null 2>&1
  for n in 32768 65536 131072 1048576; do
    k=$((200 * 65536 / n * 65536 / n)); [ $k -lt 4 ] && k=4; [ $k -gt 400 ] && k=400
    python bench.py --bodies $n --steps $k --warmup 3 --cpu-seconds 0 --no-distinct-row 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read()); f=r['roofline']
print('unroll %-3d N=%-8d %9.4f ms/step   force pass %9.4f ms   update %.4f ms' % ($u, $n, r['ms_per_step'], f['avg_launch_ms'], f['update_kernel_avg_ms']))"
  done
done
rm -f parallelnbody_amd/csrc/kernels_sym.o parallelnbody_amd/csrc/kernels_sym64.o
make -C parallelnbody_amd/csrc > /dev/null 2>&1
} > "$out" 2>&1
