# how often does an insert take far longer than its neighbours, and where does the time go?  tools/check/outliers.py [inserts]
# Three forms of the same BGKOctoMap insert (configs[1]): the cloud in pageable host memory (numpy -> hipMemcpy inside the call), in
# PINNED host memory, and already in HBM; for each the Python wall clock per call, the library's own clock (stats t_total: from the
# entry of la3dm_devmap_insert_* to its last read-back) and — gc disabled — the count of calls beyond 3 x the median.
import gc, sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, la3dm_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
xyz, origin = la3dm_amd.synthetic_scan(200000)
xyz = np.ascontiguousarray(xyz, np.float32)
pinned = torch.from_numpy(xyz).pin_memory()
d_cloud = torch.from_numpy(xyz).cuda()
gc.disable()


def report(tag, wall, lib):
    wall, lib = np.array(wall[5:]) * 1e3, np.array(lib[5:]) * 1e3
    for name, t in (("python wall", wall), ("library t_total", lib)):
        med = np.median(t)
        print(f"{tag:28s} {name:16s} median {med:.3f} ms  p99 {np.percentile(t, 99):.3f}  max {t.max():.3f}  > 3 x median: {int((t > 3 * med).sum())} of {t.size}   worst: {np.round(np.sort(t)[-4:], 2)}", flush=True)


for tag in ("cloud in pageable memory", "cloud in pinned memory", "cloud in HBM", "cloud in HBM, Python gc ON"):
    if tag.endswith("gc ON"):
        gc.enable()
    m = la3dm_amd.BGKOctoMap(**dict(la3dm_amd.BGK_YAML), device=0)
    wall, lib = [], []
    for i in range(n):
        t0 = time.perf_counter()
        if tag.startswith("cloud in HBM"):
            m.insert_pointcloud_device(d_cloud.data_ptr(), d_cloud.shape[0], origin, 0.1, 0.5, -1.0)
        elif tag == "cloud in pinned memory":
            m.insert_pointcloud(pinned.numpy(), origin, 0.1, 0.5, -1.0)
        else:
            m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0)
        wall.append(time.perf_counter() - t0)
        lib.append(m.stats()["t_total"])
    report(tag, wall, lib)
    del m

# BGKLOctoMap (the class whose insert loops showed "1 in 400 at ~10 ms" in round 5), gc off, cloud in pageable memory
gc.disable()
m = la3dm_amd.BGKLOctoMap(**dict(la3dm_amd.L_YAML), device=0)
wall, lib = [], []
for i in range(n):
    t0 = time.perf_counter()
    m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
    wall.append(time.perf_counter() - t0)
    lib.append(m.stats()["t_total"])
report("BGK-L, pageable cloud", wall, lib)
big = np.argsort(np.array(wall[5:]))[-4:] + 5
print("BGK-L: positions of the four slowest inserts", sorted(big.tolist()))
