import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
xyz, origin = la3dm_amd.synthetic_scan(200000)
for mode in ("host", "device"):
    for rep in range(4):
        m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
        if mode == "device": m.set_device_resident(True)
        t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0); t1 = time.time()
        t2 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0); t3 = time.time()
        print(mode, "insert#1 %.4f insert#2 %.4f" % (t1 - t0, t3 - t2), {k: round(v, 4) for k, v in m.stats().items() if k.startswith("t_")}, flush=True)
    t0 = time.time(); n = m.block_count(); t1 = time.time()
    print(mode, "block_count (mirror sync)", n, "%.4f s" % (t1 - t0))
