import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, la3dm_amd
from conftest import pcd_path
rows, depth = 0, 3
params = dict(la3dm_amd.L_YAML, block_depth=depth)
m = la3dm_amd.BGKLOctoMap(**params, device=0)
m.set_option("bgkl_split_rows", rows)
for i in (1, 2, 3):
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
    print("insert", i, flush=True)
    m.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
    print("  ok", flush=True)
xyz, origin = la3dm_amd.synthetic_scan(6000)
print("synthetic", flush=True)
m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
print("  ok", flush=True)
