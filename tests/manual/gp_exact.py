import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from oracle import oracle as O
params = dict(la3dm_amd.GP_YAML)
m = la3dm_amd.GPOctoMap(**params, device=0); o = O.OracleGPMap(**params)
for i in (1, 2):
    xyz, origin = la3dm_amd.load_pcd(f"tests/golden/data/sim_structured/sim_structured_{i}.pcd")
    t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.1, 8.0); t1 = time.time(); o.insert_pointcloud(xyz, origin, 0.1, 0.1, 8.0); t2 = time.time()
    a, b = m.leaves(), o.leaves()
    print("scan", i, "gpu %.3fs cpu %.3fs" % (t1 - t0, t2 - t1), "leaves", a["A"].size, "m_ivar exact", (a["A"] == b["A"]).mean(), "ivar exact", (a["B"] == b["B"]).mean(),
          "state", (a["state"] == b["state"]).mean(), "max rel dA", (np.abs(a["A"] - b["A"]) / (np.abs(b["A"]) + 1e-9)).max())
print(m.stats())
