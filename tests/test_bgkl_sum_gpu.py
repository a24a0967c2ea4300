"""BGKLOctoMap in the library's DEFAULT accumulate mode (la3dm_set_option "bgk_sum" 1, round 5): every (row, leaf) pair adds
the same fp32 k and k * label as in the ordered mode to DOUBLE sums; a neighbour's two sums are rounded to fp32 once; the
per-neighbour gate kbar > 0.001f and the fp32 update in ExtendedBlock order (src/bgkloctomap/bgkloctomap.cpp:206-231) are
unchanged.  Only the fp32 SUMMATION ORDER of include/bgkloctomap/bgklinference.h:86-87 is given up — and with it the split
tiles' scratch replay (bgkl_split_eval / kernelize / expand / add of the ordered mode: 11x the algorithmic bytes).

Checked against two oracles, in the shape of tests/test_bgk_sum_gpu.py:
  * the restatement in ITS double-sum mode (oracle.set_sum_mode(1)): same leaf structure, `classified` and states, alpha /
    beta within ONE fp32 ulp and >= 99.99 % bit-equal (the device adds the items' partial double sums of a split tile in
    item order, the restatement row by row: the double sums differ in their last bit at most, which survives the rounding
    to fp32 with probability ~2^-29);
  * the restatement in the reference's order (the default): |dp| <= 1e-5, the north-star tolerance.
The gate compares a ROUNDED sum with 0.001f: a kbar within one ulp of the threshold could in principle be gated
differently by the two modes; the check on `classified` would show it (never observed)."""
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
    assert a["block_key"].size == b["block_key"].size == c["block_key"].size, tag
    for k in ("block_key", "node_key", "classified"):
        assert (a[k] == b[k]).all(), (tag, k)
    for k in ("A", "B"):
        u = _ulps(a[k], b[k])
        assert u.max() <= 1, (tag, k, int(u.max()))
        assert (u == 0).mean() >= 0.9999, (tag, k, float((u == 0).mean()))
    far = a["state"] != b["state"]
    if far.any():      # only where the 1-ulp difference straddles a threshold
        p = _prob(a)[far]
        d = np.minimum(np.abs(p - params["free_thresh"]), np.abs(p - params["occupied_thresh"]))
        assert (d < 1e-6).all(), (tag, int(far.sum()))
    # the reference's order: same structure unless a gate or a prune decision sits on a rounding — compare where it matches
    if a["block_key"].size == c["block_key"].size and (a["node_key"] == c["node_key"]).all():
        dp = np.abs(_prob(a) - _prob(c))
        assert dp.max() <= 1e-5, (tag, float(dp.max()))
        return float(dp.max())
    raise AssertionError((tag, "leaf structure differs from the reference-order restatement"))


@pytest.fixture()
def sum64():
    from oracle import oracle as O
    yield O
    O.set_sum_mode(0)
    O.set_sum_mode(0, omp=True)


def _trio(params, O, omp=False):
    import la3dm_amd
    m = la3dm_amd.BGKLOctoMap(**params, device=0)
    m.set_option("bgk_sum", 1)
    return m, O.OracleLMap(**params, omp=omp), O.OracleLMap(**params, omp=omp)


def _insert(O, m, o64, o32, omp, *args):
    m.insert_pointcloud(*args)
    O.set_sum_mode(1, omp=omp)
    o64.insert_pointcloud(*args)
    O.set_sum_mode(0, omp=omp)
    o32.insert_pointcloud(*args)


@pytest.mark.parametrize("depth,rows", [(3, 2048), (4, 2048), (3, 0), (4, 70)])
def test_sequence_both_oracles(built, sum64, depth, rows):
    """fused sim_structured scans (pruning between them), the split path at the default threshold (no tile splits at this
    size), with every tile split (0) and with a low threshold at depth 4"""
    import la3dm_amd
    O = sum64
    params = dict(la3dm_amd.L_YAML, block_depth=depth)
    m, o64, o32 = _trio(params, O)
    m.set_option("bgkl_split_rows", rows)
    worst = 0.0
    for i in (1, 2, 3, 4):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        _insert(O, m, o64, o32, False, xyz, origin, 0.1, 0.3, 8.0)
        worst = max(worst, _check(m, o64, o32, f"d{depth} rows>{rows} scan{i}", params))
    assert worst < 1e-5


def test_synthetic_scan_with_a_sensor_block(built, sum64):
    """every beam crosses the sensor's block: its tiles take the split path at the default threshold; > 4096 tiles: the
    one-wave-per-tile form of the row kernel for the others"""
    import la3dm_amd
    O = sum64
    params = dict(la3dm_amd.L_YAML)
    m, o64, o32 = _trio(params, O, omp=True)
    xyz, origin = la3dm_amd.synthetic_scan(20000)
    _insert(O, m, o64, o32, True, xyz, origin, 0.1, 0.3, -1.0)
    assert m.stats()["n_test_blocks"] > 4096
    _check(m, o64, o32, "20k rays", params)
    _insert(O, m, o64, o32, True, xyz + np.float32(0.02), origin, 0.1, 0.3, -1.0)   # a second, fused scan (pruned blocks)
    _check(m, o64, o32, "20k rays, second scan", params)


def test_host_orchestrated_and_device_resident_agree(built):
    """the same scans through the host-orchestrated mode (la3dm_bgkl_scan_host on a packed scan) and the device-resident
    map: identical bits in the default accumulate mode too"""
    import la3dm_amd
    params = dict(la3dm_amd.L_YAML)
    maps = []
    for resident in (True, False):
        m = la3dm_amd.BGKLOctoMap(**params, device=0)
        m.set_option("bgk_sum", 1)
        m.set_device_resident(resident)
        for i in (1, 2):
            xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
            m.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
        maps.append(m.leaves())
    a, b = maps
    for k in ("block_key", "node_key", "state", "classified"):
        assert (a[k] == b[k]).all(), k
    for k in ("A", "B"):
        assert (a[k].view(np.uint32) == b[k].view(np.uint32)).all(), k


def test_default_mode_is_the_double_sum_one_and_differs_from_the_ordered_one(built, monkeypatch):
    import la3dm_amd
    monkeypatch.delenv("LA3DM_BGK_SUM", raising=False)
    params = dict(la3dm_amd.L_YAML)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    out = []
    for mode in (None, 0):
        m = la3dm_amd.BGKLOctoMap(**params, device=0)
        if mode is None:
            assert m.get_option("bgk_sum") == 1
        else:
            m.set_option("bgk_sum", mode)
        m.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
        out.append(m.leaves())
    a, b = out
    assert (a["node_key"] == b["node_key"]).all()
    assert (a["A"] != b["A"]).any()          # the default really is the other summation
    assert np.abs(_prob(a) - _prob(b)).max() <= 1e-5
