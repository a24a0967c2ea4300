import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, la3dm_amd
from conftest import pcd_path
from oracle import oracle as O
for name, (xyz, origin), mr in (("sim_structured_1", la3dm_amd.load_pcd(pcd_path("sim_structured", 1)), 8.0),
                                ("synthetic 20k rays", la3dm_amd.synthetic_scan(20000), -1.0)):
    m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
    m.insert_pointcloud(xyz, origin, 0.1, 0.3, mr)
    m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
    t0 = time.time(); m.insert_pointcloud(xyz, origin, 0.1, 0.3, mr); t1 = time.time()
    st = m.stats()
    o = O.OracleLMap(**O.L_YAML)
    t2 = time.time(); o.insert_pointcloud(xyz, origin, 0.1, 0.3, mr); t3 = time.time()
    print(name, "gpu insert %.4f s (device %.4f)" % (t1 - t0, st["t_device"]), "cpu oracle %.3f s" % (t3 - t2),
          "U", st["voxel_updates"], "P", st["pair_evals"], "rows", st["train_reads"])
