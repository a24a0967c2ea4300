import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, la3dm_amd
from conftest import pcd_path
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for rows in [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "-1,16384,4096,1024".split(","))]:
    for name, (xyz, origin), mr in (("sim_structured_1", la3dm_amd.load_pcd(pcd_path("sim_structured", 1)), 8.0),
                                    ("synthetic %d rays" % n, la3dm_amd.synthetic_scan(n), -1.0)):
        m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
        m.set_option("bgkl_split_rows", rows)
        m.insert_pointcloud(xyz, origin, 0.1, 0.3, mr)
        m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
        m.set_option("bgkl_split_rows", rows)
        t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.3, mr); t1 = time.time()
        st = m.stats()
        print("split>%d" % rows, name, "gpu insert %.4f s (device %.4f)" % (t1 - t0, st["t_device"]),
              "U", st["voxel_updates"], "P", st["pair_evals"], "rows", st["train_reads"], flush=True)
        lv = m.leaves()
        key = (name,)
        h = (lv["A"].view(np.uint32).astype(np.uint64).sum(), lv["B"].view(np.uint32).astype(np.uint64).sum(), int(lv["state"].sum()))
        print("   checksum", h)
