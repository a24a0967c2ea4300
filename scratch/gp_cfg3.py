import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from oracle import oracle as O
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rays = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
params = dict(la3dm_amd.GP_YAML, block_depth=depth)
xyz, origin = la3dm_amd.synthetic_scan(rays)
m = la3dm_amd.GPOctoMap(**params, device=0)
for rep in range(2):
    m2 = la3dm_amd.GPOctoMap(**params, device=0)
    t0 = time.time(); m2.insert_pointcloud(xyz, origin, 0.1, 0.1, -1.0); t1 = time.time()
    st = m2.stats()
    print("GPU insert %.3fs" % (t1 - t0), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st.items()})
pk = m2.packed(); print("max N", pk.c.train_max_n, "sum N^2", pk.c.train_sum_n2)
if len(sys.argv) > 3:
    o = O.OracleGPMap(**params, omp=True)
    t0 = time.time(); o.insert_pointcloud(xyz, origin, 0.1, 0.1, -1.0); t1 = time.time()
    print("CPU omp insert %.3fs" % (t1 - t0), o.stats())
    a, b = m2.leaves(), o.leaves()
    print("exact m_ivar", (a["A"] == b["A"]).mean(), "ivar", (a["B"] == b["B"]).mean(), "state", (a["state"] == b["state"]).mean())
