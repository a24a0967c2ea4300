import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from oracle import oracle as O
# BGK-L at a size where the split path carries the tiles around the sensor (threshold 4096 rows)
nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
xyz, origin = la3dm_amd.synthetic_scan(nrays)
m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
o = O.OracleLMap(**la3dm_amd.L_YAML, omp=True)
for rep in range(2):
    t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0); t1 = time.time()
    o.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0); t2 = time.time()
a, b = m.leaves(), o.leaves()
ok = a["A"].size == b["A"].size and all((a[k] == b[k]).all() for k in ("block_key", "node_key", "state", "classified")) \
    and (a["A"].view(np.uint32) == b["A"].view(np.uint32)).all() and (a["B"].view(np.uint32) == b["B"].view(np.uint32)).all()
print("bgkl", nrays, "rays device-resident", m.is_device_resident(), ": leaves", a["A"].size, "bit-identical", bool(ok), "gpu %.4f s  cpu(omp) %.2f s" % (t1 - t0, t2 - t1), flush=True)
