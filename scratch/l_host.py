import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, la3dm_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
xyz, origin = la3dm_amd.synthetic_scan(n)
for resident in (True, False):
    m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0).set_device_resident(resident)
    m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
    for rep in range(3):
        m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0).set_device_resident(resident)
        t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0); t1 = time.time()
        st = m.stats()
        print("device-resident" if m.is_device_resident() else "host-orchestrated", "insert %.4f" % (t1 - t0),
              {k: round(v, 4) for k, v in st.items() if k.startswith("t_")}, flush=True)
    lv = m.leaves()
    print("   checksum", lv["A"].view(np.uint32).astype(np.uint64).sum(), lv["B"].view(np.uint32).astype(np.uint64).sum(), int(lv["state"].sum()), lv["A"].size)
