#!/usr/bin/env python3
"""When every work item of a symmetric force launch starts and ends (reference-clock stamps left by the workgroups: contexts
created with time_kernels under NBODY_SYM_ITEM_CLOCKS=1), for the guided and the even-share plan of one system.
   python tools/even_items.py N IPT [distinct|equal]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb

n, ipt = int(sys.argv[1]), int(sys.argv[2])
posm, vel = nb.ic_plummer(n, seed=1)
if len(sys.argv) < 4 or sys.argv[3] == "distinct":
    posm[:, 3] *= np.random.default_rng(1).uniform(0.5, 1.5, n).astype(np.float32)
os.environ["NBODY_SYM_ITEM_CLOCKS"] = "1"
for even in (0, 1):
    os.environ["NBODY_SYM_EVEN"] = str(even)
    with nb.NBodyEngine(n, algorithm=2, i_per_thread=ipt, time_kernels=True) as e:
        e.set_state(posm, vel)
        e.step(0.002, 50); e.synchronize()
        L = e._L
        cnt, khz = ctypes.c_int32(), ctypes.c_int32()
        assert L.nbody_debug_sym_item_clocks(e._h, None, 0, ctypes.byref(cnt), ctypes.byref(khz)) == 0
        buf = np.zeros(2 * cnt.value, np.uint64)
        assert L.nbody_debug_sym_item_clocks(e._h, buf.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), buf.size, None, None) == 0
        items = nb.sym_plan_even(n, 256 * ipt, cnt.value)[0] if even else None
    t = buf.reshape(-1, 2).astype(np.float64) / (khz.value * 1e-3)      # microseconds
    t0 = t[:, 0].min()
    s, f = t[:, 0] - t0, t[:, 1] - t0
    d = f - s
    span = f.max()
    print(f"## N={n} bodies per lane {ipt} NBODY_SYM_EVEN={even}: {cnt.value} items, launch span {span:.1f} us, busy "
          f"{d.sum() / span:.0f} item-slots on average")
    q = [0, 10, 50, 90, 99, 100]
    print("   start  percentiles " + " ".join(f"{np.percentile(s, p):7.1f}" for p in q))
    print("   end    percentiles " + " ".join(f"{np.percentile(f, p):7.1f}" for p in q))
    print("   length percentiles " + " ".join(f"{np.percentile(d, p):7.1f}" for p in q))
    if even:
        own = (items[:, 3] & 1) != 0
        bi = 256 * ipt
        # how much of the item lies in the own block (in subtiles)
        own_sub = np.array([min(it[2], max(0, (it[0] + bi - it[1]) // 64)) if it[3] & 1 else 0 for it in items])
        for name, m in (("items that start in the own block", own), ("the others", ~own)):
            if m.any():
                print(f"   {name}: {int(m.sum())}, length mean {d[m].mean():.1f} max {d[m].max():.1f}, end mean {f[m].mean():.1f} max {f[m].max():.1f}")
        worst = np.argsort(f)[-8:][::-1]
        for w in worst:
            print(f"   late: item {w} row {items[w, 0] // bi} own subtiles {own_sub[w]} of {items[w, 2]}  start {s[w]:.1f} end {f[w]:.1f}")
