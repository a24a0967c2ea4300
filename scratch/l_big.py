import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
for rays in (50000, 200000):
    xyz, origin = la3dm_amd.synthetic_scan(rays)
    m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
    t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0); t1 = time.time()
    st = m.stats()
    print(rays, "insert %.3f s" % (t1 - t0), {k: round(st[k], 4) for k in st if k.startswith("t_")}, "rows", st["train_reads"], "P", st["pair_evals"], flush=True)
