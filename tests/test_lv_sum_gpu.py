"""BGKLVOctoMap in the library's DEFAULT accumulate mode (la3dm_set_option "bgk_sum" 1, round 5): the two running sums of a
voxel's rows (include/bgklvoctomap/bgklvinference.h:80-83, per-voxel driver src/bgklvoctomap/bgklvoctomap.cpp:176-238) are
double sums of the same fp32 kv and kv * y, rounded to fp32 once per voxel; the gate kbar > 0.001f and the LV node update
(src/bgklvoctomap/bgklvoctree_node.cpp:29-77) are unchanged.  The device's E phase adds every counted (candidate, voxel)
pair straight into the voxel's accumulators; a split cube's workgroups hand over 1 KB of partial sums each — no
[candidate][voxel] tile, no ordered add, no row scratch (bgklv_split_add_kernel, 3.2x the algorithmic bytes in round 4).

Against the restatement in ITS double-sum mode (oracle.set_sum_mode(1)): same leaf structure, `classified`, alpha / beta
within ONE fp32 ulp and >= 99.99 % bit-equal (a double sum of fp32 terms depends on the order in its last bit only);
against the restatement in the reference's order: |dp| <= 1e-5 on the LV occupancy probability (the bar of
tests/test_lv_gpu.py); alpha / beta themselves within 1e-3 relative (measured: up to 4.5e-4) (the fp32 chains of a voxel that adds thousands of rows
are that far from the correctly rounded sums; measured values are printed)."""
import numpy as np
import pytest

from conftest import pcd_path

pytestmark = pytest.mark.gpu


def _lv_prob(A, B, min_W):
    A, B = A.astype(np.float64), B.astype(np.float64)
    W = np.maximum(A + B, min_W)
    return np.where(A > B, A / (W - B) + (W - A - B) * 0.5 / (W - B), 0.5 * (W - B - A) / (W - A))


def _ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def _check(m, o64, o32, params, tag):
    a, b, c = m.leaves(), o64.leaves(), o32.leaves()
    assert a["A"].size == b["A"].size == c["A"].size, tag
    for k in ("block_key", "node_key", "classified"):
        assert (a[k] == b[k]).all(), (tag, k)
    for k in ("A", "B"):
        u = _ulps(a[k], b[k])
        assert u.max() <= 1, (tag, k, int(u.max()))
        assert (u == 0).mean() >= 0.9999, (tag, k, float((u == 0).mean()))
    assert (a["state"] == b["state"]).mean() >= 0.9999, tag
    if (a["node_key"] == c["node_key"]).all():
        # the reference's fp32 running sums carry the rounding of every partial sum: a voxel next to the sensor adds thousands
        # of rows, and its alpha / beta sit up to ~1e-4 (relative) from the correctly rounded sums this mode delivers
        worst = 0.0
        for k in ("A", "B"):
            rel = np.abs(a[k].astype(np.float64) - c[k]) / np.maximum(np.abs(c[k].astype(np.float64)), 1e-3)
            worst = max(worst, float(rel.max()))
        print(f"{tag}: max relative difference of alpha / beta from the reference order {worst:.2e}")
        assert worst <= 1e-3, (tag, worst)
        assert np.abs(_lv_prob(a["A"], a["B"], params["min_W"]) - _lv_prob(c["A"], c["B"], params["min_W"])).max() <= 1e-5, tag
    else:
        raise AssertionError((tag, "leaf structure differs from the reference-order restatement"))


@pytest.fixture()
def sum64():
    from oracle import oracle as O
    yield O
    O.set_sum_mode(0)
    O.set_sum_mode(0, omp=True)


def _insert(O, m, o64, o32, *args, omp=False):
    m.insert_pointcloud(*args)
    O.set_sum_mode(1, omp=omp)
    o64.insert_pointcloud(*args)
    O.set_sum_mode(0, omp=omp)
    o32.insert_pointcloud(*args)


@pytest.mark.parametrize("res,depth,nscan", [(0.1, 4, 4), (0.05, 5, 3), (0.1, 3, 2)])
def test_sim_unstructured_both_oracles(built, sum64, res, depth, nscan):
    import la3dm_amd
    O = sum64
    params = dict(la3dm_amd.LV_YAML, resolution=res, block_depth=depth)
    m = la3dm_amd.BGKLVOctoMap(**params, device=0)
    m.set_option("bgk_sum", 1)
    o64, o32 = O.OracleLVMap(**params), O.OracleLVMap(**params)
    for i in range(1, nscan + 1):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_unstructured", i))
        _insert(O, m, o64, o32, xyz, origin, res, 0.1, 8.0)
        _check(m, o64, o32, params, f"res{res} d{depth} scan{i}")
    assert (m.leaves()["state"] == 3).any()


def test_synthetic_scan_with_split_cubes(built, sum64):
    """8 000 rays at configs[3]'s parameters: the cubes next to the sensor are split over workgroups"""
    import la3dm_amd
    O = sum64
    params = dict(la3dm_amd.LV_YAML, resolution=0.05, block_depth=5)
    xyz, origin = la3dm_amd.synthetic_scan(8000)
    m = la3dm_amd.BGKLVOctoMap(**params, device=0)
    m.set_option("bgk_sum", 1)
    o64, o32 = O.OracleLVMap(**params, omp=True), O.OracleLVMap(**params, omp=True)   # (the OpenMP build of the restatement: equal to the serial one, tests/test_oracle.py)
    _insert(O, m, o64, o32, xyz, origin, 0.05, 0.1, 8.0, omp=True)
    _check(m, o64, o32, params, "synthetic 8 k rays")


def test_host_orchestrated_mode_and_wide_kernel(built, sum64):
    """the packed-scan entry point (la3dm_bgklv_scan_host) and a kernel of ell = five voxels (several bucket groups, most
    cubes split): the same accumulate mode, the same bounds"""
    import la3dm_amd
    O = sum64
    params = dict(la3dm_amd.LV_YAML, resolution=0.1, block_depth=4, ell=0.5)
    m = la3dm_amd.BGKLVOctoMap(**params, device=0)
    m.set_option("bgk_sum", 1)
    m.set_device_resident(False)
    o64, o32 = O.OracleLVMap(**params), O.OracleLVMap(**params)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_unstructured", 1))
    _insert(O, m, o64, o32, xyz, origin, 0.1, 0.1, 8.0)
    _check(m, o64, o32, params, "wide kernel, host-orchestrated")


def test_default_mode_is_the_double_sum_one(built, monkeypatch):
    import la3dm_amd
    monkeypatch.delenv("LA3DM_BGK_SUM", raising=False)
    params = dict(la3dm_amd.LV_YAML, resolution=0.1, block_depth=4)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_unstructured", 1))
    out = []
    for mode in (None, 0):
        m = la3dm_amd.BGKLVOctoMap(**params, device=0)
        if mode is None:
            assert m.get_option("bgk_sum") == 1
        else:
            m.set_option("bgk_sum", mode)
        m.insert_pointcloud(xyz, origin, 0.1, 0.1, 8.0)
        out.append(m.leaves())
    a, b = out
    assert (a["node_key"] == b["node_key"]).all()
    assert (a["A"] != b["A"]).any() or (a["B"] != b["B"]).any()   # the default really is the other summation
