import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
res, depth = 0.05, 5
params = dict(la3dm_amd.LV_YAML, resolution=res, block_depth=depth)
scans = [la3dm_amd.load_pcd(f"tests/golden/data/sim_unstructured/sim_unstructured_{i}.pcd") for i in range(1, 5)]
m = la3dm_amd.BGKLVOctoMap(**params, device=0)
for xyz, origin in scans:
    t1 = time.perf_counter(); m.insert_pointcloud(xyz, origin, res, 0.1, 8.0); print("insert %.6f" % (time.perf_counter() - t1)); time.sleep(0.01)
