import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, la3dm_amd, ctypes as C
md = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
mh = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0).set_device_resident(False)
print("created", md.is_device_resident(), mh.is_device_resident(), flush=True)
q = np.zeros((3, 3), np.float32)
print(mh.search_many(q), flush=True)
print(md.search_many(q), flush=True)
