import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
res, depth = 0.05, 5
params = dict(la3dm_amd.LV_YAML, resolution=res, block_depth=depth)
scans = [la3dm_amd.load_pcd(f"tests/golden/data/sim_unstructured/sim_unstructured_{i}.pcd") for i in range(1, 13)]
for mode in (True, False):
    for rep in range(2):
        m = la3dm_amd.BGKLVOctoMap(**params, device=0)
        if not mode:
            m.set_device_resident(False)
        t0 = time.perf_counter()
        ts = []
        for xyz, origin in scans:
            t1 = time.perf_counter(); m.insert_pointcloud(xyz, origin, res, 0.1, 8.0); ts.append(time.perf_counter() - t1)
        dt = time.perf_counter() - t0
    print("device-resident" if mode else "host-orchestrated", "12 scans %.4f s" % dt, "per scan ms", [round(t * 1e3, 2) for t in ts], flush=True)
    print(m.lv_stats())
