# Whole steps around the lowered AUTO threshold for the Kahan and fp64 kernels: one-sided against symmetric (forced).
for p in "f32_kahan 0.5" "f64 0"; do set -- $p; prec=$1; eps=$2
for n in 12288 16384 20480; do for a in "tiled 0" "symmetric 0" "symmetric 2"; do
set -- $a
python bench.py --bodies $n --precision $prec --eps $eps --algorithm $1 --ipt $2 --steps 300 --warmup 5 --cpu-seconds 0 --settle-seconds 0.2 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read()); d=r['config'].get('distinct_masses')
print('%-9s N=%-6d %-9s %.4f ms/step  i_per_lane %d items %d  | distinct masses %s' % ('$prec', $n, '$1', r['ms_per_step'], r['config']['i_per_lane'], r['config']['workgroups'], ('%.4f ms/step' % d['ms_per_step']) if d else '-'))"
done; done; done
