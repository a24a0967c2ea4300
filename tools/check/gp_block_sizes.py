"""Distribution of the training-block sizes N of the GP bench scans (configs[2]: 50 000 rays, 0.1 m) at a block depth,
and how many (tile, neighbour) solves fall in each size class.  gpurun -- python tools/check/gp_block_sizes.py [depth]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import la3dm_amd

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
params = dict(la3dm_amd.GP_YAML, resolution=0.1, block_depth=depth)
xyz, origin = la3dm_amd.synthetic_scan(50000)
m = la3dm_amd.GPOctoMap(**params, device=0).set_device_resident(False)
assert m.prepare(xyz, origin, 0.1, 0.5, -1.0)
pk = m.packed()
n = np.diff(pk.train_off.astype(np.int64))
print(f"depth {depth}: {n.size} training blocks, N mean {n.mean():.1f} max {n.max()}")
edges = [0, 8, 16, 24, 32, 48, 64, 80, 128, 256, 384, 1024]
h, _ = np.histogram(n, edges)
for lo, hi, c in zip(edges[:-1], edges[1:], h):
    sel = (n >= lo) & (n < hi)
    print(f"  N in [{lo:4d}, {hi:4d}): {c:6d} blocks  sum N {int(n[sel].sum()):8d}  sum N^2 {int((n[sel] ** 2).sum()):10d}")
nbr = pk.nbr.reshape(-1, 7)
mx = np.where(nbr >= 0, n[np.clip(nbr, 0, None)], 0).max(axis=1)
for t in (16, 24, 32, 48, 64):
    print(f"  test blocks whose 7 neighbours all hold <= {t} points: {int((mx <= t).sum())} of {mx.size}")
