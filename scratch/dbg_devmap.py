import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, la3dm_amd
from oracle import oracle as O
from conftest import pcd_path
params = dict(la3dm_amd.BGK_YAML)
m = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(True)
o = O.OracleMap(**params)
xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0); o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
a, b = m.leaves(), o.leaves()
for k in ("classified", "state", "A", "B"):
    d = a[k] != b[k]
    print(k, int(d.sum()))
d = a["classified"] != b["classified"]
idx = np.nonzero(d)[0][:10]
for i in idx:
    print(i, hex(a["node_key"][i]), a["classified"][i], b["classified"][i], a["state"][i], b["state"][i], a["A"][i], b["A"][i], a["B"][i], b["B"][i])
print("depths of mismatches", np.unique(a["node_key"][d] >> 16, return_counts=True))
print(np.unique(a["classified"][d]), np.unique(b["classified"][d]))
