"""Where one workgroup's tree build spends its time (bh_small_build_kernel, N <= 4096): wall_clock64 stamps at the phase
boundaries.  Needs a tuning build next to the shipped library:   make -C parallelnbody_amd/csrc variant NAME=phase_clocks EXTRA=-DNBODY_BH_PHASE_CLOCKS
    NBODY_AMD_LIB=parallelnbody_amd/libnbody_amd.phase_clocks.so
    python3 tools/bh_phases.py [N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
posm, vel = nb.ic_reference_box(n, 1000.0, seed=1)
names = ["bounds+root+thresholds", "keys", "sort", "lcp+scan", "node words", "leaves+lists", "upsweep", "hand-over"]
with nb.NBodyEngine(n, theta=1.0) as e:
    e.set_state(posm, vel)
    e.step(0.01, 50)
    acc = np.zeros(len(names))
    reps = 20
    for _ in range(reps):
        e.step(0.01, 1)
        out = (ctypes.c_longlong * (16 + 3 * 512))()
        rc = e._L.nbody_debug_bh_clocks(e._h, out)
        assert rc == 0
        c = np.array(out[:len(names) + 1], np.float64)
        acc += np.diff(c) / 100.0                     # 100 MHz -> us
    print(f"N={n}: build kernel phases, us (mean of {reps}):")
    for nm, v in zip(names, acc / reps):
        print(f"  {nm:24s} {v:8.2f}")
    if out[9] and out[10] and out[11]:
        print(f"  (inside the sort, last frame: samples ranked {(out[9] - out[2]) / 100:.2f}, buckets found and scanned {(out[10] - out[9]) / 100:.2f}, scattered {(out[11] - out[10]) / 100:.2f}, ranked {(out[3] - out[11]) / 100:.2f})")
    print(f"  {'total':24s} {acc.sum() / reps:8.2f}   {e.bh_stats()}   fullest bucket of the sample sort {out[14]} (more than 256: merge sort)")
    wg = np.array(out[16:], np.float64).reshape(512, 3)
    wg = wg[wg[:, 0] > 0][:512]      # the last frame's workgroups: start, fill end, end
    t0 = wg[:, 0].min()
    print(f"  walk kernel, all {len(wg)} workgroups of the last frame (us from the first start): starts up to {(wg[:, 0].max() - t0) / 100:.2f}, "
          f"fill {((wg[:, 1] - wg[:, 0]) / 100).mean():.2f} mean / {((wg[:, 1] - wg[:, 0]) / 100).max():.2f} max, "
          f"walk {((wg[:, 2] - wg[:, 1]) / 100).mean():.2f} mean / {((wg[:, 2] - wg[:, 1]) / 100).max():.2f} max, last end {(wg[:, 2].max() - t0) / 100:.2f}")
    print(f"  shader clock under this load: {16 * 127 * 64 / (out[15] / 100.0) / 1e3:.2f} GHz (s_sleep probe)")
