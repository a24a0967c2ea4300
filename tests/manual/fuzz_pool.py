"""Long differential fuzz of the device-resident pool against the oracle (not collected by pytest): seeded random
scenes, map parameters and scan arguments for the BGK, GP and BGK-L variants, offsets far from the origin included.
usage: python tests/manual/fuzz_pool.py [first_seed] [n_seeds] [degenerate|big|gp|likely]
`likely` (round 6): the verification configuration for holders of a real la3dm build — device options bgk_sum 0, fast_trig 3,
grid_order 1 (and gp_mode 1 for GPOctoMap) against the restatement's set_modes(1, 1) / set_gp_mode(1): Eigen 3.3.7 packet sin / cos,
pcl::VoxelGrid's own sort order, the Eigen-order GP regressor — bit for bit, all four classes, both map modes."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, la3dm_amd
from oracle import oracle as O

# The library's default accumulate mode of the BGK family (BGK, BGK-L, BGK-LV since round 5: double sums, rounded once) is
# compared with the restatement's double-sum mode within one ulp of alpha / beta (states may differ where p sits within
# 1e-6 of a threshold); LA3DM_BGK_SUM=0 in the environment runs the ordered mode, bit for bit.  GP is bit-identical in both.
LIKELY = len(sys.argv) > 3 and sys.argv[3] == "likely"
SUM1 = os.environ.get("LA3DM_BGK_SUM", "1") != "0" and not LIKELY
O.set_sum_mode(1 if SUM1 else 0)
if LIKELY:
    O.set_modes(1, 1)
    O.set_gp_mode(1)

def same(m, o, tag, ulp=False):
    a, b = m.leaves(), o.leaves()
    ok = a["A"].size == b["A"].size and all((a[k] == b[k]).all() for k in ("block_key", "node_key", "classified"))
    if ok and not ulp:
        ok = (a["state"] == b["state"]).all() and (a["A"].view(np.uint32) == b["A"].view(np.uint32)).all() and \
            (a["B"].view(np.uint32) == b["B"].view(np.uint32)).all()
    elif ok:
        for k in ("A", "B"):
            ok = ok and np.abs(a[k].view(np.int32).astype(np.int64) - b[k].view(np.int32)).max(initial=0) <= 1
        d = a["state"] != b["state"]
        if ok and d.any():
            p = b["A"][d].astype(np.float64) / (b["A"][d].astype(np.float64) + b["B"][d])
            ok = bool((np.minimum(np.abs(p - 0.3), np.abs(p - 0.7)) < 1e-6).all())
    if not ok:
        print("MISMATCH", tag, a["A"].size, b["A"].size, flush=True)
    return ok

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
degenerate = len(sys.argv) > 3 and sys.argv[3] == "degenerate"
big = len(sys.argv) > 3 and sys.argv[3] == "big"      # few seeds, large clouds, long sequences, BGK and BGK-L only
gp_heavy = len(sys.argv) > 3 and sys.argv[3] == "gp"   # GP maps with hundreds of points per block (matrix-core paths)
bad, t0 = 0, time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    res = float(rng.choice([0.05, 0.1, 0.2, 0.25]))
    depth = int(rng.choice([1, 2, 3, 4, 5]))
    kind = int(rng.integers(0, 4))
    if big:
        kind = 2 * int(rng.integers(0, 2))
    if gp_heavy:
        kind, depth = 1, int(rng.choice([3, 4]))
    common = dict(resolution=res, block_depth=depth, sf2=float(rng.choice([0.1, 1.0, 2.0])), free_thresh=0.3, occupied_thresh=0.7)
    if kind == 1:
        depth = min(depth, 4 if gp_heavy else 3)
        params = dict(common, block_depth=depth, ell=float(rng.choice([3.0, 5.0, 10.0])) * res, noise=0.01, l=100.0, min_var=0.001,
                      max_var=1000.0, max_known_var=0.02)
        m, o = la3dm_amd.GPOctoMap(**params, device=0), O.OracleGPMap(**params)
    elif kind == 3:
        depth = max(depth, 2)
        params = dict(common, block_depth=depth, ell=float(rng.choice([1.5, 2.0, 3.0])) * res, var_thresh=float(rng.choice([0.05, 0.2, 100.0])),
                      prior_A=0.001, prior_B=0.001, original_size=bool(rng.integers(0, 2)), min_W=float(rng.choice([0.1, 1.0])))
        m, o = la3dm_amd.BGKLVOctoMap(**params, device=0), O.OracleLVMap(**params)
    else:
        params = dict(common, ell=float(rng.choice([1.5, 2.0, 3.0])) * res, var_thresh=float(rng.choice([0.05, 0.15, 100.0])),
                      prior_A=0.001, prior_B=0.001)
        m, o = (la3dm_amd.BGKLOctoMap(**params, device=0), O.OracleLMap(**params)) if kind == 2 else \
               (la3dm_amd.BGKOctoMap(**params, device=0), O.OracleMap(**params))
        if kind == 2:
            m.set_option("bgkl_split_rows", int(rng.choice([0, 40, 4096])))
    if kind != 3:
        assert m.is_device_resident()
        if rng.random() < 0.25:
            m.set_device_resident(False)
    if LIKELY:
        m.set_option("bgk_sum", 0)
        m.set_option("fast_trig", 3)
        m.set_option("grid_order", 1)
        if kind == 1:
            m.set_option("gp_mode", 1)
    offset = rng.choice([0.0, 0.0, 37.3, -412.7, 5000.2]) * np.array([1, rng.choice([0, 1]), 0], np.float32)
    for scan in range(int(rng.integers(4, 9)) if big else int(rng.integers(1, 4))):
        n = int(rng.integers(1, 60 if kind == 1 else (120 if kind == 3 else 500)))
        if big:
            n = int(rng.integers(1000, 5000))
        if gp_heavy:
            n = int(rng.integers(200, 1500))
        origin = (offset + rng.uniform(-1, 1, 3)).astype(np.float32)
        pts = (origin + rng.normal(0, 1.0, (n, 3)) * rng.uniform(0.2, 3.0)).astype(np.float32)
        k = n // 4
        pts[:k] = (np.round(pts[:k] / res) * res).astype(np.float32)
        if rng.random() < 0.2:
            pts[rng.integers(0, n)] = np.nan
        if degenerate and rng.random() < 0.3:
            pts[rng.integers(0, n)] = origin                      # a hit at the sensor: zero-length beam
        if degenerate and rng.random() < 0.2:
            pts[rng.integers(0, n)] = pts[rng.integers(0, n)]     # duplicate hit
        ds = float(rng.choice([-1.0, res, 2 * res]))
        fr = float(rng.choice([0.3, 0.5, 1.0])) * max(res * 4, 0.2)
        mr = float(rng.choice([-1.0, 2.5, 6.0]))
        if kind == 3:                                             # BGK-LV: filtered clouds with a range gate
            ds, mr = res, float(rng.choice([2.5, 6.0, 8.0]))
        m.insert_pointcloud(pts, origin, ds, fr, mr)
        o.insert_pointcloud(pts, origin, ds, fr, mr)
        if not same(m, o, f"seed {seed} kind {kind} scan {scan} {params} ds={ds} fr={fr} mr={mr} offset={offset.tolist()}", ulp=SUM1 and kind in (0, 2, 3)):   # BGK, BGK-L and BGK-LV have the double-sum mode; GP does not
            bad += 1
            break
    else:
        if kind != 3:                                      # queries that run on the pool: bbox and the leaf export
            lo, hi = m.get_bbox()
            olo, ohi = o.get_bbox()
            ok = (lo == olo).all() and (hi == ohi).all()
            for state in ("occupied", "free"):
                for original in (True, False):
                    a, b = m.export_cells(state, original), o.export_cells(state, original)
                    ra = np.concatenate([a["cells"], a["rgba"], a["level"][:, None].astype(np.float32)], axis=1)
                    rb = np.concatenate([b["cells"], b["rgba"], b["level"][:, None].astype(np.float32)], axis=1)
                    ok = ok and ra.shape == rb.shape and (ra[np.lexsort(ra.T[::-1])] == rb[np.lexsort(rb.T[::-1])]).all()
            if not ok:
                print("MISMATCH (bbox / export)", f"seed {seed} kind {kind} {params}", flush=True)
                bad += 1
print(f"seeds {first}..{first + count - 1}: {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
