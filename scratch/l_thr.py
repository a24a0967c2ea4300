import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
xyz, origin = la3dm_amd.synthetic_scan(200000)
for rows in (3072, 2560, 2048, 1536):
    m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
    m.set_option("bgkl_split_rows", rows)
    ts = []
    for rep in range(5):
        t0 = time.perf_counter(); m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0); ts.append(time.perf_counter() - t0)
    print("split_rows", rows, ["%.2f" % (t * 1e3) for t in ts], flush=True)
