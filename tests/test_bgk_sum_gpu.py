"""The library's DEFAULT BGK accumulate mode (la3dm_set_option "bgk_sum" 1): every evaluated pair adds its fp32 kernel value
k (and k * y) to double accumulators in LDS, alpha / beta are rounded once.  Two kernels share the mode (bgk_kernels.h):
bgk_predict_fuse_t (round 4: per-axis distance tables, the tiles of un-pruned blocks of a scan whose labels are 0 / 1) and
bgk_predict_fuse_r (every other tile; every tile with "bgk_tables" 0) — the same pairs, the same k, the same sums.
Same pairs and the same k as the ordered kernel; only the reference's fp32 SUMMATION ORDER
(include/bgkoctomap/bgkinference.h:76-78, src/bgkoctomap/bgkoctomap.cpp:314-335) is given up.

Checked against two oracles on every BASELINE config the mode applies to:
  * the restatement in ITS double-sum mode (oracle.set_sum_mode(1)): same leaf structure, same states and `classified`,
    alpha / beta within ONE fp32 ulp (a double sum of fp32 terms depends on the order only in its last bit, which
    survives the final rounding with probability ~2^-29) and >= 99.99 % bit-equal;
  * the restatement in the reference's order (the default): same leaf structure, |dp| <= 1e-5 (the north-star
    tolerance; observed <= 5e-7), states equal except where p sits within 1e-6 of a threshold.
"""
import numpy as np
import pytest

from conftest import pcd_path

pytestmark = pytest.mark.gpu


def _ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def _prob(lv):
    a = lv["A"].astype(np.float64)
    return a / (a + lv["B"])


def _check(m, o64, o32, tag, params):
    a, b, c = m.leaves(), o64.leaves(), o32.leaves()
    for ref, name in ((b, "double-sum oracle"), (c, "reference-order oracle")):
        assert a["block_key"].size == ref["block_key"].size, (tag, name)
        assert (a["block_key"] == ref["block_key"]).all() and (a["node_key"] == ref["node_key"]).all(), (tag, name)
        assert (a["classified"] == ref["classified"]).all(), (tag, name)
    # against the double-sum restatement: one ulp, states identical
    for k in ("A", "B"):
        u = _ulps(a[k], b[k])
        assert u.max() <= 1, (tag, k, int(u.max()))
        assert (u == 0).mean() >= 0.9999, (tag, k, float((u == 0).mean()))
    far = (a["state"] != b["state"])
    if far.any():      # only where the 1-ulp difference straddles a threshold
        p = _prob(a)[far]
        d = np.minimum(np.abs(p - params["free_thresh"]), np.abs(p - params["occupied_thresh"]))
        assert (d < 1e-6).all(), (tag, int(far.sum()))
    # against the reference's summation order: the north-star tolerance
    dp = np.abs(_prob(a) - _prob(c))
    assert dp.max() <= 1e-5, (tag, float(dp.max()))
    flips = a["state"] != c["state"]
    if flips.any():
        p = _prob(c)[flips]
        d = np.minimum(np.abs(p - params["free_thresh"]), np.abs(p - params["occupied_thresh"]))
        v = np.abs(c["A"][flips] * 0 + 1)   # (variance-threshold flips are not expected at these priors)
        assert (d < 1e-6).all(), (tag, int(flips.sum()), v.size)
    return float(dp.max())


@pytest.fixture()
def oracles():
    from oracle import oracle as O
    yield O
    O.set_sum_mode(0)
    O.set_sum_mode(0, omp=True)


def _maps(la3dm_amd, O, params, omp=False, tables=1):
    m = la3dm_amd.BGKOctoMap(**params, device=0)
    m.set_option("bgk_sum", 1)
    m.set_option("bgk_tables", tables)
    assert m.is_device_resident()
    return m, O.OracleMap(**params, omp=omp), O.OracleMap(**params, omp=omp)


def _insert_both(O, o64, o32, omp, *args):
    O.set_sum_mode(1, omp=omp)
    o64.insert_pointcloud(*args)
    O.set_sum_mode(0, omp=omp)
    o32.insert_pointcloud(*args)


@pytest.mark.parametrize("tables", [1, 0])
@pytest.mark.parametrize("depth", [3, 4])
def test_config0_sim_structured_scan1(built, oracles, depth, tables):
    import la3dm_amd
    O = oracles
    params = dict(la3dm_amd.BGK_YAML, block_depth=depth)
    m, o64, o32 = _maps(la3dm_amd, O, params, tables=tables)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    _insert_both(O, o64, o32, False, xyz, origin, 0.1, 0.5, 8.0)
    _check(m, o64, o32, f"configs[0] depth {depth}", params)


@pytest.mark.parametrize("tables", [1, 0])
def test_twelve_scans_and_fifteen_reinsertions(built, oracles, tables):
    """posterior accumulation, pruning and re-testing of collapsed parents in the default mode: the 12 sim_structured
    scans fused, then scan 1 re-inserted 15 times (sim_structured_long_term) — the differences do not accumulate past
    the tolerance.  With the table kernel every scan after the first is a mix: un-pruned blocks through
    bgk_predict_fuse_t, pruned ones through bgk_predict_fuse_r behind it."""
    import la3dm_amd
    O = oracles
    params = dict(la3dm_amd.BGK_YAML)
    m, o64, o32 = _maps(la3dm_amd, O, params, tables=tables)
    for i in range(1, 13):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        _insert_both(O, o64, o32, False, xyz, origin, 0.1, 0.5, 8.0)
    _check(m, o64, o32, "12 scans", params)
    m, o64, o32 = _maps(la3dm_amd, O, params, tables=tables)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    for _ in range(15):
        m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        _insert_both(O, o64, o32, False, xyz, origin, 0.1, 0.5, 8.0)
    _check(m, o64, o32, "15 re-insertions", params)


@pytest.mark.parametrize("depth,inserts", [(3, 2), (4, 1)])
def test_config1_bgk_200k_rays(built, oracles, depth, inserts):
    """configs[1] at full size; the second insert at depth 3 meets the pruned map"""
    import la3dm_amd
    O = oracles
    params = dict(la3dm_amd.BGK_YAML, resolution=0.1, block_depth=depth)
    xyz, origin = la3dm_amd.synthetic_scan(200000)
    m, o64, o32 = _maps(la3dm_amd, O, params, omp=True)
    for _ in range(inserts):
        m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0)
        _insert_both(O, o64, o32, True, xyz, origin, 0.1, 0.5, -1.0)
    assert m.leaves()["A"].size > 1_500_000
    _check(m, o64, o32, f"configs[1] depth {depth}", params)


def test_config4_scan_1m_rays(built, oracles):
    """configs[4]'s scan (1 M rays, 0.05 m) on one GPU, default mode, against both restatements (OpenMP build)"""
    import la3dm_amd
    O = oracles
    params = dict(la3dm_amd.BGK_YAML, resolution=0.05, block_depth=3)
    xyz, origin = la3dm_amd.synthetic_scan(1000000)
    m, o64, o32 = _maps(la3dm_amd, O, params, omp=True)
    m.insert_pointcloud(xyz, origin, 0.05, 0.5, -1.0)
    _insert_both(O, o64, o32, True, xyz, origin, 0.05, 0.5, -1.0)
    assert m.leaves()["A"].size > 5_000_000
    _check(m, o64, o32, "configs[4] scan", params)


def test_training_data_with_real_valued_labels(built, oracles):
    """insert_training_data (bgkoctomap.cpp:82-212) with labels that are not 0 / 1: the kernel falls back from one
    accumulator per label to sum(k), sum(k * y), and update() runs for every test voxel (ungated)"""
    import la3dm_amd
    O = oracles
    params = dict(la3dm_amd.BGK_YAML)
    rng = np.random.default_rng(5)
    pts = rng.uniform(-2.0, 2.0, (6000, 3)).astype(np.float32)
    for labels in (rng.uniform(0.0, 1.0, 6000).astype(np.float32), (rng.uniform(0, 1, 6000) > 0.6).astype(np.float32)):
        xyzy = np.concatenate([pts, labels[:, None]], axis=1).astype(np.float32)
        m, o64, o32 = _maps(la3dm_amd, O, params)
        m.insert_training_data(xyzy)
        O.set_sum_mode(1)
        o64.insert_training_data(xyzy)
        O.set_sum_mode(0)
        o32.insert_training_data(xyzy)
        _check(m, o64, o32, "training data", params)


def test_both_modes_agree_and_default_is_the_order_free_one(built, monkeypatch):
    """la3dm_create takes the default from LA3DM_BGK_SUM (conftest pins 0 for the bit-identity suites); without it the
    default is 1; the two modes give the same leaf structure on a pruned two-scan map"""
    import la3dm_amd
    monkeypatch.delenv("LA3DM_BGK_SUM", raising=False)
    xyz, origin = la3dm_amd.synthetic_scan(20000)
    out = []
    for mode in (None, 0):
        m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
        if mode is not None:
            m.set_option("bgk_sum", mode)
        for pose in (None, (1.0, 0.5, 1.0)):
            x, o = la3dm_amd.synthetic_scan(20000, origin=pose)
            m.insert_pointcloud(x, o, 0.1, 0.5, -1.0)
        out.append(m.leaves())
    a, b = out
    assert (a["block_key"] == b["block_key"]).all() and (a["node_key"] == b["node_key"]).all()
    assert (a["A"] != b["A"]).any()          # the default really is the other kernel
    assert np.abs(_prob(a) - _prob(b)).max() <= 1e-5


@pytest.mark.parametrize("depth", [3, 4])
def test_table_kernel_and_general_kernel_agree_bit_for_bit(built, depth):
    """bgk_predict_fuse_t (default) against bgk_predict_fuse_r on the same packed scans through the C ABI (la3dm_bgk_scan_host): a
    fresh map (every block un-pruned: the table kernel alone), the third scan of a sequence (pruned blocks: both kernels in
    one call), and the same scan without LA3DM_SCAN_LABELS_01 (the general kernel alone, whatever "bgk_tables" says).
    Both kernels form the same double sums of the same fp32 terms: alpha, beta and state are equal bit for bit up to the
    order of the double additions (<= 1 ulp, >= 99.999 % equal; observed: all equal)."""
    import la3dm_amd
    params = dict(la3dm_amd.BGK_YAML, block_depth=depth)
    xyz, origin = la3dm_amd.synthetic_scan(30000)
    for earlier in (0, 2):
        m = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
        m.set_option("bgk_sum", 1)
        for s in range(earlier):
            assert m.prepare(xyz + np.float32(0.013 * (s + 1)), origin, 0.1, 0.5, -1.0)
            m.scan_host(m.packed())
            m.commit()
        assert m.prepare(xyz, origin, 0.1, 0.5, -1.0)
        pk = m.packed()
        assert pk.flags & 2          # LA3DM_SCAN_LABELS_01 from the front end
        full = int(pk.n_leaf) == int(pk.n_test_blk) * 8 ** (depth - 1)
        assert full == (earlier == 0)
        a0, b0 = pk.alpha.copy(), pk.beta.copy()
        out = []
        # (tables, flags): the general kernel; the table kernel bgk_predict_fuse_t (default); the same without the caller's
        # LA3DM_SCAN_FULL_BLOCKS (the instance with the general path compiled in); the same scan without LA3DM_SCAN_LABELS_01 (the
        # general kernel whatever the options say)
        for tables, flags in ((0, pk.flags), (1, pk.flags), (1, pk.flags & ~4), (1, pk.flags & ~2)):
            m.set_option("bgk_tables", tables)
            pk.alpha[:], pk.beta[:], pk.c.flags = a0, b0, flags
            m.scan_host(pk)
            out.append((pk.alpha.copy(), pk.beta.copy(), pk.state.copy()))
        for other in out[1:]:
            for x, y in zip(out[0][:2], other[:2]):
                u = _ulps(x, y)
                assert u.max() <= 1 and (u == 0).mean() >= 0.99999
            assert (out[0][2] == other[2]).mean() >= 0.99999
        assert (out[0][0] != a0).any()


@pytest.mark.parametrize("depth,rays", [(4, 30000), (5, 20000)])
def test_per_tile_descriptors_change_nothing(built, depth, rays):
    """block_depth >= 4 (round 6, bgk_prepare): every tile of a full block gets its own neighbour descriptor without the face
    neighbours its 4 x 4 x 4 voxel cube cannot reach (at depth 4: its own block + three of the six; at depth 5 an inner cube keeps its
    own block only).  The dropped points can reach none of the tile's leaves, so the table kernel, the general kernel (pruned blocks:
    the third scan of a sequence; unlabelled scans) and both together must give the block-wide descriptors' alpha, beta and state
    ("bgk_tile_desc" 0) — same pairs, same fp32 terms, the same double sums up to the order of the additions (<= 1 ulp; observed: equal)."""
    import la3dm_amd
    params = dict(la3dm_amd.BGK_YAML, block_depth=depth)
    xyz, origin = la3dm_amd.synthetic_scan(rays)
    for earlier in (0, 2):
        m = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
        m.set_option("bgk_sum", 1)
        for s in range(earlier):
            assert m.prepare(xyz + np.float32(0.013 * (s + 1)), origin, 0.1, 0.5, -1.0)
            m.scan_host(m.packed())
            m.commit()
        assert m.prepare(xyz, origin, 0.1, 0.5, -1.0)
        pk = m.packed()
        a0, b0 = pk.alpha.copy(), pk.beta.copy()
        out = []
        for desc, tables, flags in ((0, 1, pk.flags), (1, 1, pk.flags), (1, 0, pk.flags), (1, 1, pk.flags & ~4), (1, 1, pk.flags & ~2)):
            m.set_option("bgk_tile_desc", desc)
            m.set_option("bgk_tables", tables)
            pk.alpha[:], pk.beta[:], pk.c.flags = a0, b0, flags
            m.scan_host(pk)
            out.append((pk.alpha.copy(), pk.beta.copy(), pk.state.copy()))
        for other in out[1:]:
            for x, y in zip(out[0][:2], other[:2]):
                u = _ulps(x, y)
                assert u.max() <= 1 and (u == 0).mean() >= 0.99999
            assert (out[0][2] == other[2]).mean() >= 0.99999
        assert (out[0][0] != a0).any()
    # a kernel wider than the cube's edge (ell > 4 voxel edges): the host keeps the block-wide descriptors, results as before
    wide = dict(params, ell=0.45)
    m = la3dm_amd.BGKOctoMap(**wide, device=0).set_device_resident(False)
    m.set_option("bgk_sum", 1)
    assert m.prepare(xyz[::3], origin, 0.1, 0.5, -1.0)
    pk = m.packed()
    a0, b0 = pk.alpha.copy(), pk.beta.copy()
    res = []
    for desc in (0, 1):
        m.set_option("bgk_tile_desc", desc)
        pk.alpha[:], pk.beta[:] = a0, b0
        m.scan_host(pk)
        res.append((pk.alpha.copy(), pk.beta.copy(), pk.state.copy()))
    for x, y in zip(res[0], res[1]):
        assert (x.view(np.uint8) == y.view(np.uint8)).all()
