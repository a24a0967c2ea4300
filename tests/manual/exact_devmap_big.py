import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from oracle import oracle as O
for rays, res, depth in ((200000, 0.1, 3), (200000, 0.1, 4), (1000000, 0.05, 3)):
    params = dict(la3dm_amd.BGK_YAML, resolution=res, block_depth=depth)
    xyz, origin = la3dm_amd.synthetic_scan(rays)
    m = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(True)
    o = O.OracleMap(**params, omp=True)
    for rep in range(2):
        t0 = time.time(); m.insert_pointcloud(xyz, origin, res, 0.5, -1.0); t1 = time.time()
        o.insert_pointcloud(xyz, origin, res, 0.5, -1.0); t2 = time.time()
    a, b = m.leaves(), o.leaves()
    ok = a["A"].size == b["A"].size and all((a[k] == b[k]).all() for k in ("block_key", "node_key", "state", "classified")) \
        and (a["A"].view(np.uint32) == b["A"].view(np.uint32)).all() and (a["B"].view(np.uint32) == b["B"].view(np.uint32)).all()
    print(rays, res, depth, "leaves", a["A"].size, "bit-identical", bool(ok), "gpu %.4f s  cpu(omp) %.2f s" % (t1 - t0, t2 - t1), flush=True)
