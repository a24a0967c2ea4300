# how often does an insert take far longer than its neighbours?  tools/check/outliers.py [inserts]
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
xyz, origin = la3dm_amd.synthetic_scan(200000)
for name, cls, yaml, args in (("bgk", la3dm_amd.BGKOctoMap, la3dm_amd.BGK_YAML, (0.1, 0.5, -1.0)), ("bgkl", la3dm_amd.BGKLOctoMap, la3dm_amd.L_YAML, (0.1, 0.3, -1.0))):
    m = cls(**dict(yaml), device=0)
    if name == "bgk": m.set_device_resident(True)
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); m.insert_pointcloud(xyz, origin, *args); ts.append(time.perf_counter() - t0)
    ts = np.array(ts[3:]) * 1e3
    print(name, "median %.3f ms  max %.3f  > 3 x median: %d of %d" % (np.median(ts), ts.max(), int((ts > 3 * np.median(ts)).sum()), ts.size), np.round(np.sort(ts)[-4:], 2), flush=True)
fresh = []
for i in range(40):
    m = la3dm_amd.BGKOctoMap(**dict(la3dm_amd.BGK_YAML), device=0).set_device_resident(True)
    t0 = time.perf_counter(); m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0); fresh.append((time.perf_counter() - t0) * 1e3)
    del m
print("fresh-map first insert ms:", np.round(np.sort(np.array(fresh[1:]))[[0, len(fresh) // 2, -3, -2, -1]], 2))
