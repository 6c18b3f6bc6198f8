for n in 20480 32768 65536 131072 1048576; do for nf in 0 1; do
  k=$((200 * 65536 / n * 65536 / n)); [ $k -lt 6 ] && k=6; [ $k -gt 400 ] && k=400
  NBODY_SYM_NO_FUSE=$nf python bench.py --bodies $n --steps $k --warmup 3 --cpu-seconds 0 --settle-seconds 0.3 2>/dev/null | python -c "
import sys,json
r=json.loads(sys.stdin.read()); f=r['roofline']
print('N=%-8d %s %9.4f ms/step  %5.2f %% whole step   force pass %9.4f ms   update %.4f ms' % ($n, 'two-kernel fold' if $nf else 'fused update   ', r['ms_per_step'], r['value']*20/157.3e12*100, f['avg_launch_ms'], f['update_kernel_avg_ms']))"
done; done
