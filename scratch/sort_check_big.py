"""the ticket form of the in-house scan / sort (more than 1024 tiles): 1 M rays -> ~8 M free samples; own sort vs rocPRIM"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
n = 1000000
xyz, origin = la3dm_amd.synthetic_scan(n)
params = dict(la3dm_amd.BGK_YAML, resolution=0.05)
res = []
for own in ("1", "0"):
    os.environ["LA3DM_OWN_SORT"] = own
    m = la3dm_amd.BGKOctoMap(**params, device=0)
    ts = []
    for rep in range(3):
        t = time.time(); m.insert_pointcloud(xyz, origin, 0.05, 0.5, -1.0); ts.append(time.time() - t)
    lv = m.leaves()
    res.append((lv["A"].copy(), lv["B"].copy(), lv["state"].copy()))
    print("own" if own == "1" else "rocprim", ["%.4f" % t for t in ts], lv["A"].size, m.stats(), flush=True)
print("identical:", all((a == b).all() for a, b in zip(res[0], res[1])), flush=True)
