"""cost model of bgklv_voxel_kernel per cube (CPU, oracle front end): samples in the 27-bucket neighbourhood and in the
cube's union box, per cube of the sensor scan."""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from oracle import oracle as O
res, depth = 0.05, 5
p = dict(la3dm_amd.LV_YAML, resolution=res, block_depth=depth)
m = O.OracleLVMap(**p)
for i in (1, 6, 12):
    xyz, origin = la3dm_amd.load_pcd(f"tests/golden/data/sim_unstructured/sim_unstructured_{i}.pcd")
    xy, rays = m.training_data(xyz, origin, res, 0.1, 8.0)
    g = 4 * res
    cell = np.floor(xy[:, :3] / g).astype(np.int64)
    cmin = cell.min(0); cell -= cmin; dim = cell.max(0) + 1
    cnt = np.zeros(tuple(dim + 2), np.int64)
    np.add.at(cnt, (cell[:, 0] + 1, cell[:, 1] + 1, cell[:, 2] + 1), 1)
    # 27-neighbourhood sums
    nb = np.zeros_like(cnt)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                nb += np.roll(cnt, (dx, dy, dz), (0, 1, 2))
    v = np.sort(nb[nb > 0])[::-1]
    print(f"scan {i}: samples {len(xy)} rays {len(rays)} cubes {len(v)} stream sum {v.sum()} max {v[:8]} p99 {np.percentile(v,99):.0f} median {np.median(v):.0f}")
