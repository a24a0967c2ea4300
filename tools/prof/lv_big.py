import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
xyz, origin = la3dm_amd.synthetic_scan(n)
m = la3dm_amd.BGKLVOctoMap(**dict(la3dm_amd.LV_YAML, resolution=0.05, block_depth=5), device=0)
for rep in range(reps):
    t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.05, 0.1, 8.0); t1 = time.time()
    st = m.lv_stats()
    print("insert %.6f " % (t1 - t0), {k: (round(v, 5) if k.startswith("t_") else int(v)) for k, v in st.items()}, flush=True)
