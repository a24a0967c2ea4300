import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from oracle import oracle as O
params = dict(la3dm_amd.BGK_YAML, block_depth=3)
xyz, origin = la3dm_amd.load_pcd("tests/golden/data/sim_structured/sim_structured_1.pcd")
m = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
ok = m.prepare(xyz, origin, 0.1, 0.5, 8.0)
pk = m.packed()
import ctypes as C
from la3dm_amd import _lib
a0, b0 = pk.alpha.copy(), pk.beta.copy()
# oracle per-leaf via full insert on both and compare packed arrays after scan_host
o = O.OracleMap(**params); o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
m2 = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
m2.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
# compare unpruned: use a fresh map pair with prune disabled? simpler: compare leaves present in both by (block_key,node_key)
a, b = m2.leaves(), o.leaves()
ka = {(int(x), int(y)): i for i, (x, y) in enumerate(zip(a["block_key"], a["node_key"]))}
bad = 0; lanes = []
for j, (x, y) in enumerate(zip(b["block_key"], b["node_key"])):
    i = ka.get((int(x), int(y)))
    if i is None: continue
    if a["A"][i] != b["A"][j] or a["B"][i] != b["B"][j]:
        bad += 1
        if len(lanes) < 40: lanes.append((int(y) & 0xffff, float(a["A"][i]), float(b["A"][j]), float(a["B"][i]), float(b["B"][j])))
print("leaves gpu/oracle", a["A"].size, b["A"].size, "differing common leaves", bad)
for l in lanes: print(l)
