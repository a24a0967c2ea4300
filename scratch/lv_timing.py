import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, la3dm_amd
from conftest import pcd_path
params = dict(la3dm_amd.LV_YAML, resolution=0.05)
m = la3dm_amd.BGKLVOctoMap(**params, device=0)
tt = 0
for i in range(1, 13):
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_unstructured", i))
    t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.05, 0.1, 8.0); t1 = time.time()
    tt += t1 - t0
    st = m.lv_stats()
    print(i, "insert %.4f" % (t1 - t0), {k: round(st[k], 4) for k in st if k.startswith("t_")}, int(st["voxels"]), int(st["n_samples"]))
print("sequence total %.3f s" % tt)
