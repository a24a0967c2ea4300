import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from oracle import oracle as O
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
res = float(rng.choice([0.05, 0.1, 0.2, 0.25])); depth = int(rng.choice([1, 2, 3, 4, 5])); kind = int(rng.integers(0, 3))
common = dict(resolution=res, block_depth=depth, sf2=float(rng.choice([0.1, 1.0, 2.0])), free_thresh=0.3, occupied_thresh=0.7)
assert kind == 2
params = dict(common, ell=float(rng.choice([1.5, 2.0, 3.0])) * res, var_thresh=float(rng.choice([0.05, 0.15, 100.0])), prior_A=0.001, prior_B=0.001)
split = int(rng.choice([0, 40, 4096]))
md = la3dm_amd.BGKLOctoMap(**params, device=0); md.set_option("bgkl_split_rows", split)
mh = la3dm_amd.BGKLOctoMap(**params, device=0).set_device_resident(False); mh.set_option("bgkl_split_rows", split)
o = O.OracleLMap(**params)
offset = rng.choice([0.0, 0.0, 37.3, -412.7, 5000.2]) * np.array([1, rng.choice([0, 1]), 0], np.float32)
for scan in range(int(rng.integers(1, 4))):
    n = int(rng.integers(1, 500))
    origin = (offset + rng.uniform(-1, 1, 3)).astype(np.float32)
    pts = (origin + rng.normal(0, 1.0, (n, 3)) * rng.uniform(0.2, 3.0)).astype(np.float32)
    k = n // 4
    pts[:k] = (np.round(pts[:k] / res) * res).astype(np.float32)
    nan = False
    if rng.random() < 0.2:
        pts[rng.integers(0, n)] = np.nan; nan = True
    ds = float(rng.choice([-1.0, res, 2 * res])); fr = float(rng.choice([0.3, 0.5, 1.0])) * max(res * 4, 0.2); mr = float(rng.choice([-1.0, 2.5, 6.0]))
    for m in (md, mh, o):
        m.insert_pointcloud(pts, origin, ds, fr, mr)
    a, b, c = md.leaves(), mh.leaves(), o.leaves()
    print("scan", scan, "n", n, "nan", nan, "ds", ds, "fr", fr, "mr", mr, "split", split, "leaves dev/host/oracle", a["A"].size, b["A"].size, c["A"].size,
          "stats dev", {k: md.stats()[k] for k in ("n_hits", "n_frees", "n_test_blocks", "n_train_blocks")},
          "host", {k: mh.stats()[k] for k in ("n_hits", "n_frees", "n_test_blocks", "n_train_blocks")},
          "oracle", {k: o.stats()[k] for k in ("n_hits", "n_frees", "n_test_blocks", "n_train_blocks")}, flush=True)
# where do the maps differ?
a, c = md.leaves(), o.leaves()
ka = {(int(b), int(n)): i for i, (b, n) in enumerate(zip(a["block_key"], a["node_key"]))}
kc = {(int(b), int(n)): i for i, (b, n) in enumerate(zip(c["block_key"], c["node_key"]))}
only_a = [k for k in ka if k not in kc][:5]
only_c = [k for k in kc if k not in ka][:10]
print("only device:", [(hex(b), n >> 16, n & 0xFFFF, a["state"][ka[(b, n)]], a["A"][ka[(b, n)]], a["B"][ka[(b, n)]], a["loc"][ka[(b, n)]].tolist()) for b, n in only_a])
print("only oracle:", [(hex(b), n >> 16, n & 0xFFFF, c["state"][kc[(b, n)]], c["A"][kc[(b, n)]], c["B"][kc[(b, n)]]) for b, n in only_c])
print("origin", origin, "nan rows", np.isnan(pts).any(axis=1).nonzero()[0])
