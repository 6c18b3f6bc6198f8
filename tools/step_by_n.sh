#!/bin/bash
# Whole-step rate (bench.py's `value`: force pass + row folds + update) with the library defaults, by N and precision.
out=${1:-gpurun_out/step_by_n.txt}
{
echo "# python bench.py --bodies N --steps K --warmup 3 --cpu-seconds 0  (library defaults; K chosen for >= 0.15 s of steps)"
echo "# Plummer sphere = equal masses: the fp32 rows run the equal-mass form; the same bodies with distinct masses (general form) follow in the same row"
for prec in f32 f32_kahan f64; do
for n in 32768 65536 131072 262144 524288 1048576; do
  k=$((200 * 65536 / n * 65536 / n)); [ $k -lt 4 ] && k=4; [ $k -gt 400 ] && k=400
  eps=0; [ $prec = f32_kahan ] && eps=0.5
  python bench.py --bodies $n --precision $prec --eps $eps --steps $k --warmup 3 --cpu-seconds 0 --settle-seconds 0.3 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read()); c=r['config']; f=r['roofline']
d=c.get('distinct_masses')
print('%-9s N=%-8d %9.4f ms/step  %.4e interactions/s  %5.2f %% of peak (whole step)   force pass %9.4f ms %5.2f %%   update %.4f ms   items %d  bodies/lane %d  err %.1e  %s' % ('$prec', $n, r['ms_per_step'], r['value'], r['value']*20/(f['peak']*1e12)*100, f['avg_launch_ms'], f['frac']*100, f['update_kernel_avg_ms'], c['workgroups'], c['i_per_lane'], c['max_rel_err_sampled'], ('equal-mass form | distinct masses: %9.4f ms/step  force pass %9.4f ms %5.2f %%' % (d['ms_per_step'], d['force_pass_avg_ms'], d['roofline_frac']*100)) if d else ('equal-mass form' if c.get('equal_mass_form') else 'general form')))"
done; done
} > $out 2>&1
