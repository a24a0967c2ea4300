"""first contact with the in-house radix sort: one small insert under a watchdog"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
for n in (3000, 20000, 200000):
    xyz, origin = la3dm_amd.synthetic_scan(n)
    res = []
    for own in ("1", "0"):
        os.environ["LA3DM_OWN_SORT"] = own
        m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
        t = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0); dt = time.time() - t
        t = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0); dt2 = time.time() - t
        lv = m.leaves()
        res.append((lv["A"].copy(), lv["B"].copy(), lv["state"].copy()))
        print(n, "own" if own == "1" else "rocprim", "%.4f %.4f" % (dt, dt2), lv["A"].size, flush=True)
    print("identical:", all((a == b).all() for a, b in zip(res[0], res[1])), flush=True)
