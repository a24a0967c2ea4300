"""GPU parity tests of the device-resident map: front end, block partition,
leaf enumeration, write-back and prune all run on the GPU (include/la3dm_hip.h, la3dm_devmap_*).

Bar: every stage is integer / order work or strict-fp32 arithmetic in the reference's operation order, so the
training set, the block/leaf structure and every node (alpha, beta, state, classified) must be BIT-IDENTICAL
to the CPU oracle — no tolerance anywhere in this file.
"""
import os
import numpy as np
import pytest

from conftest import pcd_path

pytestmark = pytest.mark.gpu


def _maps(params, gp=False, omp=False):
    """device-resident map and its oracle (omp: the oracle's OpenMP build — same code, the blocks are independent —
    for the cases whose single-thread restatement takes a minute)"""
    import la3dm_amd
    from oracle import oracle as O
    if gp:
        m = la3dm_amd.GPOctoMap(**params, device=0).set_device_resident(True)
        o = O.OracleGPMap(**params, omp=omp)
    else:
        m = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(True)
        o = O.OracleMap(**params, omp=omp)
    assert m.is_device_resident()
    return m, o


def _same(m, o, tag=""):
    a, b = m.leaves(), o.leaves()
    assert a["block_key"].size == b["block_key"].size, (tag, a["block_key"].size, b["block_key"].size)
    for k in ("block_key", "node_key", "loc", "size", "classified", "state"):
        assert (a[k] == b[k]).all(), (tag, k, int((a[k] != b[k]).sum()))
    for k in ("A", "B"):
        assert (a[k].view(np.uint32) == b[k].view(np.uint32)).all(), (tag, k, int((a[k] != b[k]).sum()),
                                                                        float(np.abs(a[k] - b[k]).max()))


@pytest.mark.parametrize("case", [("sim_structured", 1, 0.1, 0.5, 8.0), ("sim_unstructured", 4, 0.1, 0.5, 8.0),
                                  ("sim_structured", 7, 0.05, 0.3, -1.0), ("sim_structured", 2, -1.0, 0.5, 6.0)])
def test_front_end_bit_identical(built, case):
    """f1: voxel grid x2, range gate, beam samples == the oracle's get_training_data, bit for bit."""
    import la3dm_amd
    from oracle import oracle as O
    ds, i, ds_res, fr, mr = case
    xyz, origin = la3dm_amd.load_pcd(pcd_path(ds, i))
    m, _ = _maps(dict(la3dm_amd.BGK_YAML))
    m.insert_pointcloud(xyz, origin, ds_res, fr, mr)
    t = m.training_data()
    ref = O.get_training_data(xyz, origin, ds_res, fr, mr)
    assert t.shape == ref.shape, (t.shape, ref.shape)
    assert (t.view(np.uint32) == ref.view(np.uint32)).all(), int((t != ref).any(axis=1).sum())


def test_front_end_synthetic_dense_origin_cell(built):
    """the voxel next to the sensor receives one sample per beam (tens of thousands of points in one cell):
    the centroid must still be the fp32 sum in cloud order"""
    import la3dm_amd
    from oracle import oracle as O
    xyz, origin = la3dm_amd.synthetic_scan(50000)
    m, _ = _maps(dict(la3dm_amd.BGK_YAML))
    m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0)
    t = m.training_data()
    ref = O.get_training_data(xyz, origin, 0.1, 0.5, -1.0)
    assert t.shape == ref.shape
    assert (t.view(np.uint32) == ref.view(np.uint32)).all()


def test_add_repeat_closed_form(built):
    """the closed-form 'add x m times' used for runs of identical samples == the sequential fp32 loop, bit for bit
    (binade crossings, ties to even, x below half an ulp, opposite signs, zeros, denormals, inf/nan)"""
    import ctypes as C
    import la3dm_amd
    from la3dm_amd import _lib
    m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
    ctx = m._M.la3dm_map_ctx(m._h)
    rng = np.random.default_rng(5)
    n = 60000
    x = (rng.uniform(0.5, 2.0, n) * 2.0 ** rng.integers(-12, 6, n)).astype(np.float32) * rng.choice([-1, 1], n).astype(np.float32)
    s = np.where(rng.random(n) < 0.3, 0.0, x * rng.uniform(-2, 3000, n) * 2.0 ** rng.integers(-2, 8, n)).astype(np.float32)
    cnt = rng.integers(0, 40000, n).astype(np.uint32)
    # hand-picked: exact ties (x = 1, 0.5, 1.5, 3 * 2^-k), zeros, denormal, inf, nan, large sums
    sp_x = np.array([1.0, 0.5, 1.5, 0.75, 1.0, -1.0, 0.0, -0.0, 1e-40, np.inf, np.nan, 1.0, 3.0, 0.1, 1e-3, 1.0], np.float32)
    sp_s = np.array([2.0 ** 24, 2.0 ** 24 + 2, 2.0 ** 24 - 7, 2.0 ** 23 + 1, 0.0, 5.0, 3.0, 0.0, 1.0, 1.0, 1.0, 2.0 ** 25,
                     2.0 ** 24 + 2, 0.0, 0.0, 16777215.0], np.float32)
    sp_m = np.array([100, 100, 100, 1000, 20000000 // 512, 30, 9, 9, 1000, 3, 3, 77, 50000, 200000, 300000, 9], np.uint32)
    x, s, cnt = np.concatenate([x, sp_x]), np.concatenate([s, sp_s]), np.concatenate([cnt, sp_m])
    f, l = np.zeros_like(x), np.zeros_like(x)
    rc = _lib.hip().la3dm_devmap_diag_add_repeat(ctx, s.ctypes.data, x.ctypes.data, cnt.ctypes.data, x.size, f.ctypes.data,
                                                 l.ctypes.data)
    assert rc == 0
    same = (f.view(np.uint32) == l.view(np.uint32)) | (np.isnan(f) & np.isnan(l))
    assert same.all(), (int((~same).sum()), s[~same][:5], x[~same][:5], cnt[~same][:5], f[~same][:5], l[~same][:5])


@pytest.mark.parametrize("depth", [3, 4])
def test_sequence_with_pruning(built, depth):
    """12 fused scans: block creation, posterior accumulation in the device pool, pruning and ragged leaf lists"""
    import la3dm_amd
    params = dict(la3dm_amd.BGK_YAML, block_depth=depth)
    m, o = _maps(params)
    for i in range(1, 13):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        st, so = m.stats(), o.stats()
        for k in ("n_hits", "n_frees", "n_bbox_blocks", "n_train_blocks", "n_test_blocks", "voxel_updates", "train_reads",
                  "pair_evals"):
            assert st[k] == so[k], (i, k, st[k], so[k])
        if i in (1, 2, 6, 12):
            _same(m, o, f"d{depth} scan{i}")
    a = m.leaves()
    assert (a["node_key"] >> 16).min() < depth - 1, "pruning must have produced coarse leaves"


def test_long_term_reinsertion(built):
    import la3dm_amd
    params = dict(la3dm_amd.BGK_YAML)
    m, o = _maps(params)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    for _ in range(15):
        m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    _same(m, o, "long_term")


def test_matches_host_orchestrated_mode(built):
    """the same map class in its two modes: identical leaves, search() and get_bbox()"""
    import la3dm_amd
    params = dict(la3dm_amd.BGK_YAML)
    md = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(True)
    mh = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
    assert md.is_device_resident() and not mh.is_device_resident()
    for i in (3, 4, 5):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_unstructured", i))
        md.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        mh.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    _same(md, mh, "modes")
    assert md.block_count() == mh.block_count()
    lo_d, hi_d = md.get_bbox()
    lo_h, hi_h = mh.get_bbox()
    assert (lo_d == lo_h).all() and (hi_d == hi_h).all()
    rng = np.random.default_rng(3)
    for p in rng.uniform(-6, 6, (200, 3)):
        assert md.search(*p) == mh.search(*p)


def test_raycaster_on_device_resident_map(built):
    """RayCaster over the lazily mirrored pool: voxel walk, keys and node copies == the oracle's"""
    import la3dm_amd
    params = dict(la3dm_amd.BGK_YAML, block_depth=4)
    m, o = _maps(params)
    for i in (1, 2):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    rng = np.random.default_rng(2)
    lv = o.leaves()
    for _ in range(40):
        s3 = lv["loc"][rng.integers(0, lv["loc"].shape[0])] + rng.uniform(-0.04, 0.04, 3).astype(np.float32)
        e3 = s3 + rng.uniform(-4, 4, 3).astype(np.float32)
        a, b = m.raycast(s3, e3), o.raycast(s3, e3)
        assert a["p"].shape == b["p"].shape
        for k in ("p", "block_key", "node_key", "valid"):
            assert (a[k] == b[k]).all(), k
        v = a["valid"].astype(bool)
        for k in ("A", "B", "state"):
            assert (a[k][v] == b[k][v]).all(), k


def test_batched_search_on_the_device_pool(built):
    """search() for many points answered from the device pool == the host map's search (incl. pruned regions, voxel
    faces and missing blocks), before and after the mirror exists"""
    import la3dm_amd
    params = dict(la3dm_amd.BGK_YAML)
    md = la3dm_amd.BGKOctoMap(**params, device=0)
    mh = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
    assert md.search_many(np.zeros((3, 3), np.float32))["exists"].sum() == 0          # empty map
    for i in (1, 2, 3, 4):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        md.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        mh.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    rng = np.random.default_rng(8)
    q = np.concatenate([rng.uniform(-3, 12, (20000, 3)), np.round(rng.uniform(-2, 10, (5000, 3)) * 10) / 10,
                        [[500, 500, 500]]]).astype(np.float32)
    a, b = md.search_many(q), mh.search_many(q)
    assert md.is_device_resident()
    for k in ("exists", "state"):
        assert (a[k] == b[k]).all(), k
    for k in ("A", "B"):
        assert (a[k].view(np.uint32) == b[k].view(np.uint32)).all(), k
    assert 0 < int(a["exists"].sum()) < q.shape[0]
    for p in q[:50]:
        e, A, B, s = mh.search(*map(float, p))
        j = int(np.nonzero((q == p).all(1))[0][0])
        assert bool(a["exists"][j]) == e and a["A"][j] == np.float32(A) and a["state"][j] == s


def test_synthetic_scan(built):
    import la3dm_amd
    xyz, origin = la3dm_amd.synthetic_scan(30000)
    for depth in (3, 4):
        params = dict(la3dm_amd.BGK_YAML, block_depth=depth)
        m, o = _maps(params, omp=True)
        for _ in range(2):
            m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0)
            o.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0)
        _same(m, o, f"synth d{depth}")


def test_growing_map_moving_sensor(built):
    """a sensor moving through a large scene: the block pool is re-allocated and the hash table rebuilt several
    times while earlier blocks keep being updated (overlapping scans)"""
    import la3dm_amd
    params = dict(la3dm_amd.BGK_YAML)
    m, o = _maps(params)
    xyz, origin = la3dm_amd.synthetic_scan(20000)
    origin = np.asarray(origin, np.float32)
    nb = []
    for k, off in enumerate([(0, 0, 0), (6.0, 0, 0), (12.0, 3.0, 0), (3.0, 0.5, 0), (18.0, -4.0, 0.2), (24.0, 0, 0)]):
        off = np.array(off, np.float32)
        m.insert_pointcloud(xyz + off, origin + off, 0.1, 0.5, -1.0)
        o.insert_pointcloud(xyz + off, origin + off, 0.1, 0.5, -1.0)
        nb.append(m.stats()["n_test_blocks"])
    _same(m, o, "moving")
    assert m.block_count() > 60000      # > 2 pool re-allocations (x1.5 growth) and a table rebuild (> 16 k blocks)


def test_edge_cases(built):
    import la3dm_amd
    params = dict(la3dm_amd.BGK_YAML)
    m, o = _maps(params)
    m.insert_pointcloud(np.zeros((0, 3), np.float32), [0, 0, 0], 0.1, 0.5, 8.0)
    m.insert_pointcloud(np.array([[20, 0, 0]], np.float32), [0, 0, 0], 0.1, 0.5, 8.0)   # beyond max_range
    assert m.block_count() == 0
    pts = np.array([[0.2, 0.2, 0.2], [0.2, 0.0, 0.0], [0.6, 0.6, 0.6], [1.0, 1.0, 1.0], [-0.2, 0.1, 0.1],
                    [np.nan, 0.0, 0.0]], np.float32)
    m.insert_pointcloud(pts[:5], [0, 0, 0], -1.0, 0.5, -1.0)     # points on block faces / corners
    o.insert_pointcloud(pts[:5], [0, 0, 0], -1.0, 0.5, -1.0)
    _same(m, o, "faces")
    m.insert_pointcloud(np.repeat(pts, 5, axis=0), [0.05, 0, 0], 0.1, 0.3, -1.0)   # duplicates + a NaN point
    o.insert_pointcloud(np.repeat(pts, 5, axis=0), [0.05, 0, 0], 0.1, 0.3, -1.0)
    _same(m, o, "dups")
    # an empty labelled set is a no-op on the pool; the split prepare()/commit() form is host-orchestrated: the call
    # moves the map out of the device-resident mode (one download of the pool) and keeps the content
    before = m.leaves()
    m.insert_training_data(np.zeros((0, 4), np.float32))
    assert m.is_device_resident()
    m.prepare(np.zeros((0, 3), np.float32), [0, 0, 0], 0.1, 0.5, 8.0)
    assert not m.is_device_resident()
    after = m.leaves()
    assert all((before[k] == after[k]).all() for k in before)


def test_gp_variant(built):
    """GPOctoMap on the device-resident pool (same f1-f3 stages, GP regression kernels in the middle)"""
    import la3dm_amd
    params = dict(la3dm_amd.GP_YAML)
    m, o = _maps(params, gp=True)
    for i in (1, 2):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        m.insert_pointcloud(xyz[::3], origin, 0.1, 0.5, 8.0)
        o.insert_pointcloud(xyz[::3], origin, 0.1, 0.5, 8.0)
    _same(m, o, "gp")


def test_randomised_small_scenes(built):
    """seeded random scenes and map parameters (resolution, depth, length scale, sampling, range gate, voxel filter
    on/off): both map modes must reproduce the oracle bit for bit over three fused scans each"""
    import la3dm_amd
    from oracle import oracle as O
    rng = np.random.default_rng(2026)
    for case in range(6):   # (round 6: 12 -> 6 cases, suite time; tests/manual/fuzz_pool.py is the long form)
        res = float(rng.choice([0.05, 0.1, 0.2]))
        depth = int(rng.choice([1, 2, 3, 4]))
        params = dict(resolution=res, block_depth=depth, sf2=float(rng.choice([0.1, 1.0])),
                      ell=float(rng.choice([1.5, 2.0, 3.0])) * res, free_thresh=0.3, occupied_thresh=0.7,
                      var_thresh=float(rng.choice([0.05, 100.0])), prior_A=0.001, prior_B=0.001)
        md = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(True)
        mh = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
        o = O.OracleMap(**params, omp=True)
        for scan in range(3):
            n = int(rng.integers(1, 400))
            origin = rng.uniform(-1, 1, 3).astype(np.float32)
            # points on a few planes / blobs around the sensor, some exactly on voxel and block faces
            pts = origin + rng.normal(0, 1.0, (n, 3)).astype(np.float32) * rng.uniform(0.2, 3.0)
            k = n // 4
            pts[:k] = np.round(pts[:k] / res) * res
            ds = float(rng.choice([-1.0, res, 2 * res]))
            fr = float(rng.choice([0.3, 0.5, 1.0])) * max(res * 4, 0.2)
            mr = float(rng.choice([-1.0, 2.5, 6.0]))
            for m in (md, mh, o):
                m.insert_pointcloud(pts, origin, ds, fr, mr)
            _same(md, o, f"case{case} scan{scan} device-resident {params} ds={ds} fr={fr} mr={mr}")
            _same(mh, o, f"case{case} scan{scan} host-orchestrated")


@pytest.mark.parametrize("x0", [4096.2, 1000.2, 50000.2])
def test_float_stepped_candidate_list_repeats_and_gaps(built, x0):
    """get_blocks_in_bbox steps floats: far from the origin a block index repeats (the serial reference then processes
    that test block twice, the second time on the first pass's posterior) or is skipped (its points stay
    geometrically visible to the neighbours' R-tree queries but train nothing).  x0 values found by replaying the
    loop: 4096.2 and 50000.2 repeat an index, 1000.2 skips one.  All three implementations must agree bit for bit."""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.BGK_YAML)
    rng = np.random.default_rng(7)
    n = 6000
    pts = np.stack([np.float32(x0) + rng.uniform(0, 12.0, n), rng.uniform(-2, 2, n), rng.uniform(0, 2, n)], 1).astype(np.float32)
    pts[0] = (np.float32(x0), 0, 1)
    pts[1] = (np.float32(x0) + np.float32(12.0), 0, 1)
    origin = np.array([x0 + 6.0, 0.0, 1.0], np.float32)
    md = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(True)
    mh = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
    o = O.OracleMap(**params)
    for m in (md, mh, o):
        m.insert_pointcloud(pts, origin, -1.0, 0.5, -1.0)
        m.insert_pointcloud(pts[::2], origin, 0.1, 0.5, -1.0)
    _same(mh, o, "host-orchestrated")
    _same(md, o, "device-resident")


def test_voxel_grid_index_overflow_passthrough_and_depth5(built):
    """(a) a cloud whose voxel-grid index space exceeds int32: pcl::VoxelGrid hands the input back unfiltered (both
    filter calls); (b) block_depth 5 (4096 leaves per block, the deepest the device-resident map supports)"""
    import la3dm_amd
    from oracle import oracle as O
    rng = np.random.default_rng(17)
    params = dict(la3dm_amd.BGK_YAML)
    m, o = _maps(params, omp=True)
    pts = (rng.uniform(-1, 1, (150, 3)) * np.array([55.0, 55.0, 15.0])).astype(np.float32)   # 2200 x 2200 x 600 cells of 5 cm
    origin = np.zeros(3, np.float32)
    m.insert_pointcloud(pts, origin, 0.05, 2.0, -1.0)
    o.insert_pointcloud(pts, origin, 0.05, 2.0, -1.0)
    t, ref = m.training_data(), O.get_training_data(pts, origin, 0.05, 2.0, -1.0)
    assert t.shape == ref.shape and (t.view(np.uint32) == ref.view(np.uint32)).all()
    assert (t[:, 3] == 1).sum() == 150                      # nothing was merged: the filter passed the cloud through
    _same(m, o, "passthrough")
    params = dict(la3dm_amd.BGK_YAML, block_depth=5, resolution=0.05)
    m, o = _maps(params, omp=True)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 3))
    for _ in range(2):
        m.insert_pointcloud(xyz[::5], origin, 0.05, 0.5, 5.0)
        o.insert_pointcloud(xyz[::5], origin, 0.05, 0.5, 5.0)
    _same(m, o, "depth5")


def _sorted_cells(e):
    rows = np.concatenate([e["cells"], e["rgba"], e["level"][:, None].astype(np.float32)], axis=1)
    return rows[np.lexsort(rows.T[::-1])]


@pytest.mark.parametrize("depth,gp", [(3, False), (4, False), (3, True)])
def test_leaf_export_on_the_device_pool(built, depth, gp):
    """f3 leaf export: the static node's publish loop (occupied cells coloured by height, free cells by
    probability, collapsed leaves kept or expanded) runs on the pool — no mirror refresh — and equals the oracle's
    restatement cell for cell; get_bbox comes from the pool's key box."""
    import la3dm_amd
    params = dict(la3dm_amd.GP_YAML if gp else la3dm_amd.BGK_YAML, block_depth=depth)
    m, o = _maps(params, gp=gp)
    assert m.export_cells("occupied")["cells"].shape == (0, 4)                # empty pool
    for i in (1, 2, 3, 4):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        m.insert_pointcloud(xyz if not gp else xyz[::3], origin, 0.1, 0.5, 8.0)
        o.insert_pointcloud(xyz if not gp else xyz[::3], origin, 0.1, 0.5, 8.0)
    lo, hi = m.get_bbox()
    olo, ohi = o.get_bbox()
    assert (lo == olo).all() and (hi == ohi).all()
    n = 0
    for state in ("occupied", "free"):
        for original in (True, False):
            for zr in ((0.0, 0.0), (-0.3, 1.7)):
                a, b = m.export_cells(state, original, *zr), o.export_cells(state, original, *zr)
                assert a["cells"].shape == b["cells"].shape, (state, original, zr, a["cells"].shape, b["cells"].shape)
                assert (_sorted_cells(a) == _sorted_cells(b)).all(), (state, original, zr)
                n += a["cells"].shape[0]
    assert n > 10000
    occ = m.export_cells("occupied")
    if not gp:
        assert occ["level"].max() >= 1                                        # collapsed (coarse) occupied leaves
    # the host-orchestrated mode walks its blocks on the CPU: same cells
    h = (la3dm_amd.GPOctoMap if gp else la3dm_amd.BGKOctoMap)(**params, device=0).set_device_resident(False)
    for i in (1, 2, 3, 4):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        h.insert_pointcloud(xyz if not gp else xyz[::3], origin, 0.1, 0.5, 8.0)
    for state in ("occupied", "free"):
        assert (_sorted_cells(h.export_cells(state, False)) == _sorted_cells(m.export_cells(state, False))).all()


@pytest.mark.parametrize("depth", [3, 4])
def test_insert_training_data_on_the_pool(built, depth):
    """BGKOctoMap::insert_training_data (labelled points instead of a scan, updates not gated on kbar) on the
    device-resident pool == the oracle's restatement == the host-orchestrated mode, bit for bit; scans and labelled
    sets interleave on the same map"""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.BGK_YAML, block_depth=depth)
    m, o = _maps(params)
    h = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    xy = O.get_training_data(xyz, origin, 0.1, 0.5, 8.0)            # a realistic labelled set: scan 1's training data
    for mm in (m, o, h):
        mm.insert_training_data(xy)
    assert m.is_device_resident() and not h.is_device_resident()
    _same(m, o, "training data")
    _same(h, o, "training data, host mode")
    lv = m.leaves()
    assert lv["classified"].mean() > 0.95       # ungated: every leaf of a test block (collapsed parents lose the flag)
    xyz2, origin2 = la3dm_amd.load_pcd(pcd_path("sim_structured", 2))
    for mm in (m, o, h):
        mm.insert_pointcloud(xyz2, origin2, 0.1, 0.5, 8.0)
    pts = np.array([[0.3, 0.2, 0.1, 1.0], [0.35, 0.2, 0.1, 0.0], [5.0, 5.0, 1.0, 1.0]], np.float32)
    for mm in (m, o, h):
        mm.insert_training_data(pts)
        mm.insert_training_data(np.zeros((0, 4), np.float32))
    _same(m, o, "interleaved")
    _same(h, o, "interleaved, host mode")


def test_cloud_already_in_hbm(built):
    """insert_pointcloud_device: the cloud is a device buffer (here a torch tensor); same map as the host-cloud call"""
    import torch
    import la3dm_amd
    params = dict(la3dm_amd.BGK_YAML)
    a, o = _maps(params)
    for i in (1, 2, 3):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        d = torch.from_numpy(np.ascontiguousarray(xyz, np.float32)).to("cuda:0")
        torch.cuda.synchronize()
        a.insert_pointcloud_device(d.data_ptr(), d.shape[0], origin, 0.1, 0.5, 8.0)
        o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        st, so = a.stats(), o.stats()
        assert st["n_hits"] == so["n_hits"] and st["voxel_updates"] == so["voxel_updates"]
    _same(a, o, "device cloud")
    h = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
    with pytest.raises(RuntimeError, match="not device resident"):
        h.insert_pointcloud_device(d.data_ptr(), d.shape[0], origin, 0.1, 0.5, 8.0)


def test_config5_scan_at_full_size(built):
    """BASELINE configs[4]'s scan (1 M rays, 0.05 m) on one GPU, device-resident: every leaf bit-identical to the
    oracle (its OpenMP build: the restatement is the same code, the blocks are independent)"""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.BGK_YAML, resolution=0.05)
    xyz, origin = la3dm_amd.synthetic_scan(1000000)
    m = la3dm_amd.BGKOctoMap(**params, device=0)
    assert m.is_device_resident()
    o = O.OracleMap(**params, omp=True)
    m.insert_pointcloud(xyz, origin, 0.05, 0.5, -1.0)
    o.insert_pointcloud(xyz, origin, 0.05, 0.5, -1.0)
    st, so = m.stats(), o.stats()
    for k in ("n_hits", "n_frees", "n_train_blocks", "n_test_blocks", "voxel_updates", "pair_evals", "train_reads"):
        assert st[k] == so[k], (k, st[k], so[k])
    assert st["voxel_updates"] > 15_000_000
    _same(m, o, "1M rays @ 0.05 m")


def test_unstructured_sequence_all_variants_of_the_pool(built):
    """the other data set of the reference (data/sim_unstructured, 12 scans) through the three pool variants — BGK at
    0.1 m, GP on every third point, BGK-L — each against its oracle restatement"""
    import la3dm_amd
    from oracle import oracle as O
    cases = [(la3dm_amd.BGKOctoMap, O.OracleMap, dict(la3dm_amd.BGK_YAML), 1, 0.5),
             (la3dm_amd.GPOctoMap, O.OracleGPMap, dict(la3dm_amd.GP_YAML), 3, 0.5),
             (la3dm_amd.BGKLOctoMap, O.OracleLMap, dict(la3dm_amd.L_YAML), 1, 0.3)]
    for cls, ocls, params, step, fr in cases:
        m, o = cls(**params, device=0), ocls(**params)
        assert m.is_device_resident()
        for i in range(1, 13):
            xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_unstructured", i))
            m.insert_pointcloud(xyz[::step], origin, 0.1, fr, 8.0)
            o.insert_pointcloud(xyz[::step], origin, 0.1, fr, 8.0)
        _same(m, o, cls.__name__)
        assert m.leaves()["A"].size > 20000


@pytest.mark.parametrize("variant", ["bgk", "gp", "bgkl"])
def test_nan_in_the_first_point_of_an_unfiltered_cloud(built, variant):
    """ds_resolution < 0 and a NaN in the first hit: it seeds the reference's bbox reduction (`<` never replaces a NaN),
    the limits are NaN, get_blocks_in_bbox makes no step and the scan is a no-op — in the oracle, on the pool and in the
    host-orchestrated mode (which used to size a buffer from the NaN range).  A NaN anywhere else is ignored.
    Found by tests/manual/fuzz_pool.py."""
    import la3dm_amd
    from oracle import oracle as O
    cls, ocls, params = {"bgk": (la3dm_amd.BGKOctoMap, O.OracleMap, la3dm_amd.BGK_YAML),
                         "gp": (la3dm_amd.GPOctoMap, O.OracleGPMap, la3dm_amd.GP_YAML),
                         "bgkl": (la3dm_amd.BGKLOctoMap, O.OracleLMap, la3dm_amd.L_YAML)}[variant]
    pts = np.array([[np.nan, 0.0, 0.0], [1.0, 0.2, 0.1], [0.3, 0.9, 0.2], [-0.8, 0.1, 0.3], [0.4, -0.7, 0.6]], np.float32)
    origin = np.array([0.05, 0.05, 0.05], np.float32)
    for resident in (True, False):
        m, o = cls(**params, device=0).set_device_resident(resident), ocls(**params)
        m.insert_pointcloud(pts, origin, -1.0, 0.5, -1.0)
        o.insert_pointcloud(pts, origin, -1.0, 0.5, -1.0)
        assert m.block_count() == 0 and o.leaves()["A"].size == 0, (variant, resident)
        later = np.ascontiguousarray(pts[[1, 0, 2, 3, 4]])                      # the NaN is no longer first: ignored
        m.insert_pointcloud(later, origin, -1.0, 0.5, -1.0)
        o.insert_pointcloud(later, origin, -1.0, 0.5, -1.0)
        assert m.block_count() > 0
        _same(m, o, f"{variant} resident={resident}")
        m.insert_pointcloud(pts, origin, -1.0, 0.5, -1.0)                       # and a no-op again on a filled map
        o.insert_pointcloud(pts, origin, -1.0, 0.5, -1.0)
        _same(m, o, f"{variant} resident={resident} after the no-op")


_SWITCH_SCRIPT = r"""
import sys, os, zlib
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, la3dm_amd
from conftest import pcd_path
m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
for i in (1, 2):
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
    m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
lv = m.leaves()
print("CHK", lv["A"].size, zlib.crc32(lv["A"].tobytes()), zlib.crc32(lv["B"].tobytes()), zlib.crc32(lv["state"].tobytes()),
      zlib.crc32(lv["block_key"].tobytes()))
"""


def test_cloud_filter_sorts_on_the_digits_the_last_insert_needed(built):
    """round 5: the cloud's own voxel filter sorts its cell keys on as many 8-bit digits as the previous insert of the map
    needed and runs again on all four when the grid that comes back is larger (devmap.hip voxel_grid): a tiny scan (< 2^16
    cells), the same scan at full extent (< 2^24), a finer grid over a stretched copy (> 2^24), then small again — the
    training set of every insert equals the restatement's bit for bit, and so does the map at the end"""
    import la3dm_amd
    from oracle import oracle as O
    xyz, origin = la3dm_amd.synthetic_scan(12000)
    origin = np.asarray(origin, np.float32)
    m, o = _maps(dict(la3dm_amd.BGK_YAML), omp=True)
    cases = [(0.05, 0.1), (1.0, 0.1), (3.0, 0.04), (0.05, 0.1), (1.0, 0.1)]
    cells = []
    for k, (scale, ds) in enumerate(cases):
        pts = ((xyz - origin) * np.float32(scale) + origin).astype(np.float32)
        ext = np.floor(pts.max(0) / ds) - np.floor(pts.min(0) / ds) + 1
        cells.append(float(np.prod(ext)))
        m.insert_pointcloud(pts, origin, ds, 0.5, -1.0)
        o.insert_pointcloud(pts, origin, ds, 0.5, -1.0)
        t = m.training_data()
        ref = O.get_training_data(pts, origin, ds, 0.5, -1.0)
        assert t.shape == ref.shape, (k, t.shape, ref.shape)
        assert (t.view(np.uint32) == ref.view(np.uint32)).all(), (k, int((t != ref).any(axis=1).sum()))
    assert cells[0] < 2 ** 16 < cells[1] < 2 ** 24 < cells[2], cells   # the sequence crosses both digit boundaries, both ways
    _same(m, o, "digits")


@pytest.mark.parametrize("switch", ["LA3DM_MAILBOX", "LA3DM_PUBLISH_IN_KERNEL", "LA3DM_OWN_SORT", "LA3DM_TEST_SORT", "LA3DM_FORCE_SLAB=1", "LA3DM_DEPTH3"])
def test_fallback_switches_give_the_same_map(built, switch):
    """the A/B switches of the front end (copy + sync read-backs, a publish launch per read-back instead of the producing
    kernel's own mailbox write, rocPRIM's sort instead of devmap_sort.h) read their environment once per process: a child
    process per mode, two fused scans, every leaf equal to the default mode's"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(extra):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", _SWITCH_SCRIPT, root], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return [l for l in r.stdout.splitlines() if l.startswith("CHK")][-1]

    # (LA3DM_FORCE_SLAB=1, round 6: the x-slab partition of the sharded insert — per-cell counts first, pairs / sort / CSR / neighbour tables
    #  inside the pass — on an unsharded map, its "slab" being the whole test list: the same map, leaf for leaf;
    #  LA3DM_DEPTH3=0: the general leaf-list / write-back + prune kernels instead of the block_depth-3 forms of devmap_depth3.h)
    name, _, value = switch.partition("=")
    assert run({name: value or "0"}) == run({})


@pytest.mark.parametrize("cls", ["BGKOctoMap", "GPOctoMap", "BGKLOctoMap", "BGKLVOctoMap"])
def test_insert_after_set_resolution_and_set_block_depth(built, cls):
    """set_resolution / set_block_depth (reference src/bgkoctomap/bgkoctomap.cpp:66-80 and the GP / BGK-L / BGK-LV twins)
    rebuild the LUT, the device context and the pool: an insert afterwards is bit-identical to a map constructed with
    those values, in both map modes; once the map holds blocks the calls raise."""
    import la3dm_amd
    yaml = {"BGKOctoMap": la3dm_amd.BGK_YAML, "GPOctoMap": la3dm_amd.GP_YAML, "BGKLOctoMap": la3dm_amd.L_YAML,
            "BGKLVOctoMap": la3dm_amd.LV_YAML}[cls]
    K = getattr(la3dm_amd, cls)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    res, depth, fr = (0.1, 4, 0.5) if cls != "BGKLVOctoMap" else (0.1, 4, 0.3)
    for resident in (True, False):
        a = K(**dict(yaml, resolution=0.2, block_depth=3), device=0)
        a.set_device_resident(resident)
        a.set_resolution(res).set_block_depth(depth)
        assert a.is_device_resident() == resident
        b = K(**dict(yaml, resolution=res, block_depth=depth), device=0)
        b.set_device_resident(resident)
        for m in (a, b):
            m.insert_pointcloud(xyz, origin, res, fr, 8.0)
        la, lb = a.leaves(), b.leaves()
        assert la["A"].size == lb["A"].size > 1000
        for k in ("block_key", "node_key", "state", "classified"):
            assert (la[k] == lb[k]).all(), (cls, resident, k)
        for k in ("A", "B"):
            assert (la[k].view(np.uint32) == lb[k].view(np.uint32)).all(), (cls, resident, k)
        with pytest.raises(RuntimeError):
            a.set_block_depth(3)
        with pytest.raises(RuntimeError):
            a.set_resolution(0.3)


def test_look_back_error_path_leaves_the_process_usable(built, monkeypatch):
    """a prefix-sum / radix launch whose bounded look-back spin trips flags an error bit, the host poisons
    the map at its next counter read-back and every later call on it fails loudly — a defensible decision that had no test.
    LA3DM_INJECT_SCAN_STUCK = n (read once, at la3dm_devmap_create) makes the n-th counter read-back of a map report that
    bit.  Here: the insert fails with an error that names the cause, the map stays failed (no silent half-inserted state is
    served), destroying it works, and the NEXT map created in the same process — same context type, same stream machinery,
    same arenas' allocator — inserts the same scan bit-identically to the oracle."""
    import la3dm_amd
    from oracle import oracle as O
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    params = dict(la3dm_amd.BGK_YAML)
    monkeypatch.setenv("LA3DM_INJECT_SCAN_STUCK", "2")
    bad = la3dm_amd.BGKOctoMap(**params, device=0)
    assert bad.is_device_resident()
    with pytest.raises(RuntimeError) as e:
        bad.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    assert "prefix-sum launch found its state in use" in str(e.value)
    with pytest.raises(RuntimeError):                      # poisoned: a second insert is refused, not attempted
        bad.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    del bad
    monkeypatch.delenv("LA3DM_INJECT_SCAN_STUCK")
    m = la3dm_amd.BGKOctoMap(**params, device=0)
    m.set_option("bgk_sum", 0)
    o = O.OracleMap(**params)
    for i in (1, 2):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    a, b = m.leaves(), o.leaves()
    assert (a["block_key"] == b["block_key"]).all() and (a["node_key"] == b["node_key"]).all()
    assert (a["A"].view(np.uint32) == b["A"].view(np.uint32)).all() and (a["B"].view(np.uint32) == b["B"].view(np.uint32)).all()
    assert (a["state"] == b["state"]).all()


def test_counter_block_survives_non_insert_entry_points(built):
    """an insert leaves the counter block clean for the next one (no dm_begin launch) — any other entry point that
    runs a scan or a sort writes counter slots in between and must hand the next insert a reset block.  insert, then
    diag_scan / diag_sort / export_cells, then insert == the same two inserts back to back, bit for bit."""
    import ctypes as C
    import la3dm_amd
    from la3dm_amd import _lib
    H = _lib.hip()
    owner = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)      # (its context; the raw pools below are this test's own)
    clouds = [la3dm_amd.load_pcd(pcd_path("sim_structured", i)) for i in (1, 2)]

    def run(disturb):
        dm = C.c_void_p()
        assert H.la3dm_devmap_create(owner.ctx(), C.byref(dm)) == 0
        try:
            for k, (xyz, origin) in enumerate(clouds):
                xyz = np.ascontiguousarray(xyz, np.float32)
                org = np.asarray(origin, np.float32)
                assert H.la3dm_devmap_insert_pointcloud_host(dm, xyz.ctypes.data, xyz.shape[0], 3, org.ctypes.data, 0.1, 0.5, 8.0, None) == 0
                if disturb and k == 0:
                    x = np.arange(100000, dtype=np.uint32) % 7
                    out, aux = np.zeros_like(x), np.zeros(4, np.uint32)
                    assert H.la3dm_devmap_diag_scan(dm, 0, x.ctypes.data, x.size, out.ctypes.data, aux.ctypes.data) == 0
                    keys = (x * 2654435761 % 4096).astype(np.uint32)
                    ko, vo = np.zeros_like(keys), np.zeros_like(keys)
                    assert H.la3dm_devmap_diag_sort(dm, keys.ctypes.data, x.ctypes.data, x.size, 12, ko.ctypes.data, vo.ctypes.data) == 0
                    assert (np.diff(ko.astype(np.int64)) >= 0).all()
            nb, npb = C.c_uint32(), C.c_uint32()
            assert H.la3dm_devmap_block_count(dm, C.byref(nb), C.byref(npb)) == 0
            n = nb.value * npb.value
            keys, A, B, S = np.zeros(nb.value, np.int64), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.uint8)
            assert H.la3dm_devmap_download(dm, keys.ctypes.data, A.ctypes.data, B.ctypes.data, S.ctypes.data) == 0
            order = np.argsort(keys, kind="stable")       # (pool slots are handed out by an atomic: their order differs from run to run)
            per = lambda a: a.reshape(nb.value, npb.value)[order]
            return keys[order], per(A), per(B), per(S)
        finally:
            H.la3dm_devmap_destroy(dm)

    ref, got = run(False), run(True)
    assert ref[0].size > 100
    for a, b in zip(ref, got):
        assert a.shape == b.shape and (a.view(np.uint8) == b.view(np.uint8)).all()
    # the class's own export (an exclusive scan on the pool) between two inserts
    m, o = _maps(dict(la3dm_amd.BGK_YAML))
    for k, (xyz, origin) in enumerate(clouds):
        m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        if k == 0:
            assert m.export_cells("occupied")["cells"].shape[0] > 0
    _same(m, o, "insert / export / insert")
