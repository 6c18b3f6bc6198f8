# Whole steps, one box: one-sided against symmetric pass around the AUTO threshold; Plummer sphere (equal masses -> both
# kernels' equal-mass forms) and, after the bar, the same bodies with distinct masses (general forms).
for n in 12288 16384 20480 24576 28672 32768; do for a in tiled symmetric; do
python bench.py --bodies $n --algorithm $a --steps 300 --warmup 5 --cpu-seconds 0 --settle-seconds 0.2 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read()); d=r['config'].get('distinct_masses')
print('N=%-6d %-9s %.4f ms/step  %.2f %% (whole step)  i_per_lane %d items %d  | distinct masses %s' % ($n, '$a', r['ms_per_step'], r['value']*20/157.3e12*100, r['config']['i_per_lane'], r['config']['workgroups'], ('%.4f ms/step' % d['ms_per_step']) if d else '-'))"
done; done
