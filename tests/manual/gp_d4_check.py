import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, la3dm_amd
from conftest import pcd_path
from oracle import oracle as O
params = dict(la3dm_amd.GP_YAML, block_depth=4)
m = la3dm_amd.GPOctoMap(**params, device=0)
o = O.OracleGPMap(**params, omp=True)
xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
xyz = xyz[::2]
t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.1, 8.0); t1 = time.time()
o.insert_pointcloud(xyz, origin, 0.1, 0.1, 8.0); t2 = time.time()
a, b = m.leaves(), o.leaves()
pk_stats = m.stats()
print("gpu %.3f s cpu(omp) %.3f s" % (t1 - t0, t2 - t1), "leaves", a["A"].size, "train blocks", pk_stats["n_train_blocks"])
for k in ("A", "B", "state", "classified"):
    d = a[k] != b[k]
    print(k, "mismatches", int(d.sum()), "max abs", float(np.abs(a[k].astype(np.float64) - b[k]).max()))
