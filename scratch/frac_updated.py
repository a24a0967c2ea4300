import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
xyz, origin = la3dm_amd.synthetic_scan(200000)
m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0)
lv = m.leaves()
print("leaves", lv["A"].size, "classified", int(lv["classified"].sum()), "fraction %.3f" % lv["classified"].mean(), "stats U", m.stats()["voxel_updates"])
