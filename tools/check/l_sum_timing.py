"""BGKLOctoMap insert_pointcloud_device of the synthetic 200 k-ray scan in both accumulate modes: per-insert times (the bench's
bgkl leg measures the same).  gpurun -- python tools/check/l_sum_timing.py [rays]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import la3dm_amd

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
xyz, origin = la3dm_amd.synthetic_scan(rays)
for mode in (0, 1):
    ts = []
    for rep in range(4):
        m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
        m.set_option("bgk_sum", mode)
        t0 = time.time()
        m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
        ts.append(time.time() - t0)
        st = m.stats()
        dev = st["t_device"]
    lv = m.leaves()
    print(f"bgk_sum {mode}: insert {np.median(ts[1:]) * 1e3:.2f} ms (last device {dev * 1e3:.2f} ms)  leaves {lv['A'].size} "
          f"checksum {int(lv['A'].view(np.uint32).astype(np.uint64).sum())}", flush=True)
