import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
xyz, origin = la3dm_amd.synthetic_scan(200000)
m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
for _ in range(3):
    m2 = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
    t0 = time.time(); m2.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0); t1 = time.time()
    print("insert %.4f" % (t1 - t0), {k: round(v, 4) for k, v in m2.stats().items() if k.startswith("t_")})
