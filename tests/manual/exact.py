import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from oracle import oracle as O
for depth in (3, 4):
    params = dict(la3dm_amd.BGK_YAML, block_depth=depth)
    m = la3dm_amd.BGKOctoMap(**params, device=0); o = O.OracleMap(**params)
    for i in range(1, 13):
        xyz, origin = la3dm_amd.load_pcd(f"tests/golden/data/sim_structured/sim_structured_{i}.pcd")
        m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0); o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    a, b = m.leaves(), o.leaves()
    print("depth", depth, "leaves", a["A"].size, "A exact:", (a["A"] == b["A"]).mean(), "B exact:", (a["B"] == b["B"]).mean(),
          "state:", (a["state"] == b["state"]).mean(), "classified:", (a["classified"] == b["classified"]).mean(),
          "max|dA|", np.abs(a["A"]-b["A"]).max(), "max|dB|", np.abs(a["B"]-b["B"]).max())
xyz, origin = la3dm_amd.synthetic_scan(200000)
m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0); o = O.OracleMap(**la3dm_amd.BGK_YAML)
m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0); o.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0)
a, b = m.leaves(), o.leaves()
print("synthetic 200k: leaves", a["A"].size, "A exact:", (a["A"] == b["A"]).mean(), "B exact:", (a["B"] == b["B"]).mean(),
      "state:", (a["state"] == b["state"]).mean(), "max|dA|", np.abs(a["A"]-b["A"]).max(), "max|dB|", np.abs(a["B"]-b["B"]).max())
print(m.stats())
