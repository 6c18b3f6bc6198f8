for n in 8192 10240 12288 14336 16384 18432; do for a in "tiled 0" "symmetric 0" "symmetric 2"; do
set -- $a
python bench.py --bodies $n --algorithm $1 --ipt $2 --steps 400 --warmup 5 --cpu-seconds 0 --settle-seconds 0.2 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read()); d=r['config'].get('distinct_masses')
print('N=%-6d %-9s %.4f ms/step  %.2f %% (whole step)  i_per_lane %d items %d  | distinct masses %s' % ($n, '$1', r['ms_per_step'], r['value']*20/157.3e12*100, r['config']['i_per_lane'], r['config']['workgroups'], ('%.4f ms/step' % d['ms_per_step']) if d else '-'))"
done; done
