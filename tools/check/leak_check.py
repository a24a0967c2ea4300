import sys, os, gc
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, la3dm_amd, torch
from conftest import pcd_path
xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
def free():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
f0 = None
for it in range(60):
    for cls, params in ((la3dm_amd.BGKOctoMap, la3dm_amd.BGK_YAML), (la3dm_amd.GPOctoMap, la3dm_amd.GP_YAML),
                        (la3dm_amd.BGKLOctoMap, la3dm_amd.L_YAML), (la3dm_amd.BGKLVOctoMap, la3dm_amd.LV_YAML)):
        m = cls(**params, device=0)
        m.insert_pointcloud(xyz[::4], origin, 0.1, 0.5, 8.0)
        if it % 2 and cls in (la3dm_amd.BGKOctoMap, la3dm_amd.GPOctoMap):
            m.set_device_resident(False)
        n = m.leaves()["A"].size
        del m
    gc.collect()
    if it == 4: f0 = free()
f1 = free()
print("free after warm-up %.1f MB, at the end %.1f MB, drift %.2f MB" % (f0 / 2**20, f1 / 2**20, (f0 - f1) / 2**20))
