import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
xyz, origin = la3dm_amd.synthetic_scan(n)
m = la3dm_amd.BGKLOctoMap(**dict(la3dm_amd.L_YAML, resolution=0.1, block_depth=3), device=0)
for rep in range(reps):
    t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0); t1 = time.time()
    st = m.stats()
    print("insert %.6f  U %d rows %d" % (t1 - t0, st["voxel_updates"], st["train_reads"]), flush=True)
