import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
res = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
xyz, origin = la3dm_amd.synthetic_scan(n)
m = la3dm_amd.BGKOctoMap(**dict(la3dm_amd.BGK_YAML, resolution=res), device=0).set_device_resident(True)
for rep in range(reps):
    t0 = time.time(); m.insert_pointcloud(xyz, origin, res, 0.5, -1.0); t1 = time.time()
    print("insert %.6f" % (t1 - t0), flush=True)
