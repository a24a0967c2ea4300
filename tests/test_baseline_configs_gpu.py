"""Full-size parity tests of the BASELINE.json configs: the HIP path against the CPU oracle's
OpenMP build (same restatement, blocks are independent) at the sizes BASELINE.md §4 names — not a subsample.

  configs[1]  BGKOctoMap, synthetic 200 000-ray scan, 0.1 m, bgkoctomap.yaml, block_depth 3 (YAML) and 4 (ctor default)
              reference path: src/bgkoctomap/bgkoctomap.cpp:214-366, include/bgkoctomap/bgkinference.h:73-126
  configs[2]  GPOctoMap, synthetic 50 000-ray scan, 0.1 m, gpoctomap.yaml (free_resolution 0.1, depth 3)
              reference path: src/gpoctomap/gpoctomap.cpp:205-350, include/gpoctomap/gpregressor.h:42-92
  configs[3]  BGKLVOctoMap, all 12 sim_unstructured scans, 0.05 m, bgklvoctomap.yaml (depth 5, max_range 8)
              reference path: src/bgklvoctomap/bgklvoctomap.cpp:89-285

Bar: block/leaf structure, state, `classified` and every (alpha, beta) / (m_ivar, ivar) BIT-IDENTICAL to the oracle
(north-star tolerance 1e-5 on the occupancy probability is therefore met with margin 0)."""
import numpy as np
import pytest

from conftest import pcd_path

pytestmark = pytest.mark.gpu


def _bit_identical(a, b, tag):
    assert a["block_key"].size == b["block_key"].size, (tag, a["block_key"].size, b["block_key"].size)
    for k in ("block_key", "node_key", "state", "classified"):
        assert (a[k] == b[k]).all(), (tag, k, int((a[k] != b[k]).sum()))
    for k in ("A", "B"):
        bad = a[k].view(np.uint32) != b[k].view(np.uint32)
        assert not bad.any(), (tag, k, int(bad.sum()), float(np.abs(a[k] - b[k]).max()))


@pytest.mark.parametrize("depth,inserts", [(3, 2), (4, 1)])
def test_config1_bgk_200k_rays(built, depth, inserts):
    """configs[1] at full size on the device-resident map; the second insert at depth 3 runs on the pruned map."""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.BGK_YAML, resolution=0.1, block_depth=depth)
    xyz, origin = la3dm_amd.synthetic_scan(200000)
    m = la3dm_amd.BGKOctoMap(**params, device=0)
    assert m.is_device_resident()
    o = O.OracleMap(**params, omp=True)
    for rep in range(inserts):
        m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0)
        o.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0)
    a, b = m.leaves(), o.leaves()
    assert a["A"].size > 1_500_000
    _bit_identical(a, b, f"configs[1] depth {depth}")
    pa = a["A"].astype(np.float64) / (a["A"].astype(np.float64) + a["B"])
    pb = b["A"].astype(np.float64) / (b["A"].astype(np.float64) + b["B"])
    assert np.abs(pa - pb).max() <= 1e-5          # the north-star tolerance, written out


def test_config1_bgk_200k_rays_host_orchestrated_kernel_only(built):
    """the same scan through the split prepare()/la3dm_bgk_scan_host()/commit() form (the C-ABI hot path on its own)"""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.BGK_YAML, resolution=0.1, block_depth=3)
    xyz, origin = la3dm_amd.synthetic_scan(200000)
    m = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
    o = O.OracleMap(**params, omp=True)
    m.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0)
    o.insert_pointcloud(xyz, origin, 0.1, 0.5, -1.0)
    _bit_identical(m.leaves(), o.leaves(), "configs[1] host-orchestrated")


def test_config2_gp_50k_rays(built):
    """configs[2] exactly: GPOctoMap(**gpoctomap.yaml), synthetic_scan(50000), ds 0.1, free_resolution 0.1"""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.GP_YAML)
    assert params["block_depth"] == 3 and params["resolution"] == 0.1
    xyz, origin = la3dm_amd.synthetic_scan(50000)
    m = la3dm_amd.GPOctoMap(**params, device=0)
    assert m.is_device_resident()
    o = O.OracleGPMap(**params, omp=True)
    m.insert_pointcloud(xyz, origin, 0.1, 0.1, -1.0)
    o.insert_pointcloud(xyz, origin, 0.1, 0.1, -1.0)
    a, b = m.leaves(), o.leaves()
    assert a["A"].size > 500_000
    _bit_identical(a, b, "configs[2]")
    # tolerance of the north star on the logistic occupancy probability (gpoctree_node.cpp:31-34)
    max_ivar = 1.0 / params["min_var"]
    pa = 1.0 / (1.0 + np.exp(-params["l"] * a["A"].astype(np.float64) / max_ivar))
    pb = 1.0 / (1.0 + np.exp(-params["l"] * b["A"].astype(np.float64) / max_ivar))
    assert np.abs(pa - pb).max() <= 1e-5
    st = m.stats()
    assert st["voxel_updates"] == o.stats()["voxel_updates"]


def test_config3_lv_full_sequence(built):
    """configs[3]: the 12 sim_unstructured scans fused at 0.05 m (block_depth 5); training samples and segments are
    compared after every scan, all leaves after scans 1, 4, 8 and 12."""
    import la3dm_amd
    from oracle import oracle as O
    res, depth = 0.05, 5
    params = dict(la3dm_amd.LV_YAML, resolution=res, block_depth=depth)
    m = la3dm_amd.BGKLVOctoMap(**params, device=0)
    o = O.OracleLVMap(**params)
    for i in range(1, 13):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_unstructured", i))
        m.insert_pointcloud(xyz, origin, res, 0.1, 8.0)
        o.insert_pointcloud(xyz, origin, res, 0.1, 8.0)
        s, r = m.lv_training()
        s2, r2 = o.training_data(xyz, origin, res, 0.1, 8.0)
        assert s.shape == s2.shape and (s == s2).all() and (r == r2).all(), i
        if i in (1, 4, 8, 12):
            a, b = m.leaves(), o.leaves()
            assert (a["loc"] == b["loc"]).all() and (a["size"] == b["size"]).all(), i
            _bit_identical(a, b, f"configs[3] scan {i}")
    assert a["A"].size > 1_000_000
    assert (a["state"] == 3).any()                            # UNCERTAIN voxels exist
    assert ((a["node_key"] >> 28) < depth - 1).any()          # pruning collapsed some groups


def test_config2_gp_50k_rays_depth4(built):
    """VERDICT r03 item 3 — the block sizes bench.py's `gp.depth4` leg quotes its MFMA roofline on: configs[2]'s 50 000-ray scan
    at block_depth 4 (the constructor default; training blocks of up to ~500 points, i.e. the blocked Cholesky / TRSM on the
    matrix cores), every leaf against the OpenMP build of the restatement.  Round 6 (suite time: the restatement's scalar solve of the
    full scan is 80 s of 128 threads): the rays of the scan's +x +y quadrant around the sensor — the same point density, hence the
    same block sizes, a quarter of the blocks; the same sample `gp.depth4.cpu_baseline` times."""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.GP_YAML, block_depth=4)
    xyz, origin = la3dm_amd.synthetic_scan(50000)
    d = xyz - np.asarray(origin, np.float32)[None, :]
    xyz = np.ascontiguousarray(xyz[(d[:, 0] >= 0) & (d[:, 1] >= 0)])
    m = la3dm_amd.GPOctoMap(**params, device=0)
    o = O.OracleGPMap(**params, omp=True)
    m.insert_pointcloud(xyz, origin, 0.1, 0.1, -1.0)
    o.insert_pointcloud(xyz, origin, 0.1, 0.1, -1.0)
    a, b = m.leaves(), o.leaves()
    assert a["A"].size > 250_000
    _bit_identical(a, b, "configs[2] at depth 4 (one quadrant)")
    assert m.stats()["voxel_updates"] == o.stats()["voxel_updates"]


def test_config3_lv_synthetic_50k_rays(built):
    """VERDICT r03 item 3 — the size bench.py's `lv.synthetic_50k` leg quotes: BGKLVOctoMap at configs[3]'s parameters
    (0.05 m, block_depth 5, max_range 8) on the synthetic 50 000-ray scan, inserted twice (the second insert meets the
    first one's pruned / classified nodes).  The restatement's O(hits^2) ray shortening and its voxel loop run in the
    OpenMP build (hits / blocks in parallel, the same values in the same order: tests/test_oracle.py checks that build
    against the serial one).  Training set (samples, segments) and every leaf."""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.LV_YAML, resolution=0.05, block_depth=5)
    xyz, origin = la3dm_amd.synthetic_scan(50000)
    m = la3dm_amd.BGKLVOctoMap(**params, device=0)
    o = O.OracleLVMap(**params, omp=True)
    for k in range(2):
        m.insert_pointcloud(xyz, origin, 0.05, 0.1, 8.0)
        o.insert_pointcloud(xyz, origin, 0.05, 0.1, 8.0)
        if k == 0:
            xy_ref, rays_ref = o.training_data(xyz, origin, 0.05, 0.1, 8.0)
            xy, rays = m.lv_training()
            assert xy.shape == xy_ref.shape and (xy.view(np.uint32) == xy_ref.view(np.uint32)).all()
            assert rays.shape == rays_ref.shape and (rays.view(np.uint32) == rays_ref.view(np.uint32)).all()
        a, b = m.leaves(), o.leaves()
        assert (a["loc"] == b["loc"]).all() and (a["size"] == b["size"]).all(), k
        _bit_identical(a, b, f"BGK-LV synthetic 50 k rays, insert {k + 1}")
    assert m.lv_stats()["n_samples"] > 1_000_000
