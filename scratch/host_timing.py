import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
xyz, origin = la3dm_amd.synthetic_scan(200000)
for rep in range(3):
    m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0).set_device_resident(False)
    t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0); t1 = time.time()
    print("host-mode insert %.4f" % (t1 - t0), {k: round(v, 4) for k, v in m.stats().items() if k.startswith("t_")}, flush=True)
m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0); t1 = time.time()
print("bgkl insert %.4f" % (t1 - t0), {k: round(v, 4) for k, v in m.stats().items() if k.startswith("t_")}, flush=True)
