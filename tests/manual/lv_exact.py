import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from oracle import oracle as O
res = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nscan = int(sys.argv[3]) if len(sys.argv) > 3 else 3
params = dict(la3dm_amd.LV_YAML, resolution=res, block_depth=depth)
m = la3dm_amd.BGKLVOctoMap(**params, device=0); o = O.OracleLVMap(**params)
for i in range(1, nscan + 1):
    xyz, origin = la3dm_amd.load_pcd(f"tests/golden/data/sim_unstructured/sim_unstructured_{i}.pcd")
    t0 = time.time(); m.insert_pointcloud(xyz, origin, res, 0.1, 8.0); t1 = time.time(); o.insert_pointcloud(xyz, origin, res, 0.1, 8.0); t2 = time.time()
    a, b = m.leaves(), o.leaves()
    same = a["A"].size == b["A"].size and (a["block_key"] == b["block_key"]).all() and (a["node_key"] == b["node_key"]).all()
    print("scan", i, "gpu %.3fs cpu %.3fs" % (t1 - t0, t2 - t1), "leaves", a["A"].size, b["A"].size, "struct", same)
    if same:
        print("   A exact", (a["A"] == b["A"]).mean(), "B exact", (a["B"] == b["B"]).mean(), "state", (a["state"] == b["state"]).mean(),
              "classified", (a["classified"] == b["classified"]).mean(), "max|dA|", np.abs(a["A"] - b["A"]).max(), "max|dB|", np.abs(a["B"] - b["B"]).max())
print(m.lv_stats()); print(o.stats())
