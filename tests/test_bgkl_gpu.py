"""GPU parity tests of BGKLOctoMap: block-level BGK with free-space line segments, in the
default device-resident mode (the whole insert_pointcloud on the GPU) and in the host-orchestrated mode.

HIP path (la3dm_bgkl_scan_* through the map class) vs the CPU oracle restatement of
src/bgkloctomap/bgkloctomap.cpp + include/bgkloctomap/bgklinference.h.  The sums run in row order and sin/cos are
correctly rounded on both sides, so alpha, beta, state and `classified` must be bit-identical (no tolerance;
the north-star bar is |dp| <= 1e-5).  Parity unpinned against the reference itself (Eigen/PCL absent).
"""
import numpy as np
import pytest

from conftest import pcd_path

pytestmark = pytest.mark.gpu


def _maps(params, resident=True):
    import la3dm_amd
    from oracle import oracle as O
    m = la3dm_amd.BGKLOctoMap(**params, device=0)
    assert m.is_device_resident()                      # the default: front end, rows, partition, prune on the GPU too
    m.set_device_resident(resident)
    return m, O.OracleLMap(**params)


def _same(m, o, tag=""):
    a, b = m.leaves(), o.leaves()
    assert a["block_key"].size == b["block_key"].size, (tag, a["block_key"].size, b["block_key"].size)
    for k in ("block_key", "node_key", "loc", "size", "classified", "state"):
        assert (a[k] == b[k]).all(), (tag, k, int((a[k] != b[k]).sum()))
    for k in ("A", "B"):
        d = a[k].view(np.uint32) != b[k].view(np.uint32)
        assert not d.any(), (tag, k, int(d.sum()), float(np.abs(a[k] - b[k]).max()))


def test_segment_distance_and_kernel_primitives(built):
    """point_to_line_dist cases (degenerate, before / after / inside the segment) == oracle"""
    from oracle import oracle as O
    L = O.lib()
    p = np.array([0.3, 0.2, 0.1], np.float32)
    cases = [((0, 0, 0), (1, 0, 0)), ((1, 0, 0), (2, 0, 0)), ((-2, 0, 0), (-1, 0, 0)), ((0.3, 0.2, 0.1), (0.3, 0.2, 0.1)),
             ((0, 0, 0), (0.00001, 0, 0))]
    exp = [np.hypot(0.2, 0.1), np.sqrt(0.7 ** 2 + 0.05), np.sqrt(1.3 ** 2 + 0.05), 0.0, np.sqrt(0.09 + 0.05)]
    for (a, b), e in zip(cases, exp):
        d = L.orc_l_seg_dist(p, np.array(a, np.float32), np.array(b, np.float32))
        assert abs(d - e) < 1e-6, (a, b, d, e)


@pytest.mark.parametrize("depth", [3, 4])
def test_sequence(built, depth):
    """fused scans with bgkloctomap.yaml parameters (free_resolution 0.3): training rows, posterior, pruning"""
    import la3dm_amd
    params = dict(la3dm_amd.L_YAML, block_depth=depth)
    m, o = _maps(params)
    for i in range(1, 7):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        m.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
        o.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
        st, so = m.stats(), o.stats()
        for k in ("n_hits", "n_frees", "n_bbox_blocks", "n_train_blocks", "n_test_blocks", "voxel_updates", "pair_evals",
                  "train_reads"):
            assert st[k] == so[k], (i, k, st[k], so[k])
        _same(m, o, f"d{depth} scan{i}")
    assert (m.leaves()["classified"] == 1).sum() > 1000


def test_unstructured_and_bypass(built):
    import la3dm_amd
    params = dict(la3dm_amd.L_YAML)
    m, o = _maps(params)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_unstructured", 2))
    m.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
    o.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
    _same(m, o, "unstructured")
    m.insert_pointcloud(xyz[:800], origin, -1.0, 0.5, 6.0)      # no voxel grid, range gate
    o.insert_pointcloud(xyz[:800], origin, -1.0, 0.5, 6.0)
    _same(m, o, "bypass")


def test_edge_cases(built):
    import la3dm_amd
    params = dict(la3dm_amd.L_YAML)
    m, o = _maps(params)
    m.insert_pointcloud(np.zeros((0, 3), np.float32), [0, 0, 0], 0.1, 0.3, 8.0)
    m.insert_pointcloud(np.array([[20, 0, 0]], np.float32), [0, 0, 0], 0.1, 0.3, 8.0)
    assert m.block_count() == 0
    pts = np.array([[0.2, 0.2, 0.2], [0.25, 0.0, 0.0], [0.6, 0.6, 0.6], [1.0, 1.0, 1.0], [-0.2, 0.1, 0.1]], np.float32)
    m.insert_pointcloud(pts, [0.05, 0, 0], -1.0, 0.3, -1.0)      # hits closer than free_resolution: degenerate beams
    o.insert_pointcloud(pts, [0.05, 0, 0], -1.0, 0.3, -1.0)
    _same(m, o, "short beams")
    m.insert_pointcloud(np.repeat(pts, 3, axis=0), [0, 0, 0.5], -1.0, 0.2, -1.0)
    o.insert_pointcloud(np.repeat(pts, 3, axis=0), [0, 0, 0.5], -1.0, 0.2, -1.0)
    _same(m, o, "dups")


@pytest.mark.parametrize("rows,depth,dense", [(0, 3, 1), (0, 4, 1), (300, 3, 1), (70, 4, 1), (0, 3, 0), (70, 4, 0)])
def test_split_tiles(built, rows, depth, dense):
    """tiles with more than `bgkl_split_rows` rows run the split path (distance test / kernel evaluation spread over
    waves, ordered replay of the sums): same bits as the row-serial kernel and the oracle, whatever the threshold and
    in both forms of the replay (rows expanded for all items at once + copy-only replay / expansion inside the replay)"""
    import la3dm_amd
    params = dict(la3dm_amd.L_YAML, block_depth=depth)
    m, o = _maps(params)
    m.set_option("bgkl_split_rows", rows)
    m.set_option("bgkl_dense_add", dense)
    for i in (1, 2, 3):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        m.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
        o.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
        _same(m, o, f"split rows>{rows} d{depth} scan{i}")
    xyz, origin = la3dm_amd.synthetic_scan(6000)                 # every beam crosses the sensor's block
    m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
    o.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
    _same(m, o, f"split rows>{rows} d{depth} synthetic")
    m.insert_pointcloud(np.zeros((0, 3), np.float32), origin, 0.1, 0.3, 8.0)
    _same(m, o, "empty after split")


def test_many_tiles_one_wave_per_tile(built):
    """above 4096 tiles the row-serial kernel runs one wave per tile (throughput form) instead of eight (latency form):
    a scan large enough to take that launch, with the tiles around the sensor on the split path"""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.L_YAML)
    m = la3dm_amd.BGKLOctoMap(**params, device=0)
    o = O.OracleLMap(**params, omp=True)
    xyz, origin = la3dm_amd.synthetic_scan(20000)
    m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
    o.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
    assert m.stats()["n_test_blocks"] > 4096
    _same(m, o, "20k rays")


@pytest.mark.parametrize("depth", [3, 4])
def test_host_orchestrated_mode(built, depth):
    """the same scans through the host-orchestrated path (host front end / partition / rows, la3dm_bgkl_scan_host):
    same bits as the oracle, hence as the device-resident mode"""
    import la3dm_amd
    params = dict(la3dm_amd.L_YAML, block_depth=depth)
    m, o = _maps(params, resident=False)
    assert not m.is_device_resident()
    for i in (1, 2, 3):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        m.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
        o.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
        st, so = m.stats(), o.stats()
        for k in ("n_hits", "n_frees", "n_train_blocks", "n_test_blocks", "voxel_updates", "pair_evals", "train_reads"):
            assert st[k] == so[k], (i, k, st[k], so[k])
    _same(m, o, f"host mode d{depth}")
    xyz, origin = la3dm_amd.synthetic_scan(6000)
    m.set_option("bgkl_split_rows", 300)
    m.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
    o.insert_pointcloud(xyz, origin, 0.1, 0.3, -1.0)
    _same(m, o, "host mode synthetic")


@pytest.mark.parametrize("resident", [True, False])
@pytest.mark.parametrize("split_rows", [0, 4096])
def test_beam_built_from_a_nan_point(built, resident, split_rows):
    """an unfiltered cloud (ds_resolution < 0) with a NaN point: its beam still owns a finite origin sample, so the
    origin's block gets a row whose segment is NaN; the reference's dense kernel matrix then holds NaN (the `< 0 -> 0`
    clean-up lets it through) and poisons ybar / kbar of every leaf that meets the row — the kbar > 0.001 gate rejects
    those updates.  Found by tests/manual/fuzz_pool.py."""
    import la3dm_amd
    params = dict(la3dm_amd.L_YAML, block_depth=2, resolution=0.2, ell=0.3)
    pts = np.array([[1.0, 0.2, 0.1], [np.nan, 0.0, 0.0], [0.3, 0.9, 0.2], [-0.8, 0.1, 0.3]], np.float32)
    origin = np.array([0.05, 0.05, 0.05], np.float32)
    m, o = _maps(params, resident=resident)
    m.set_option("bgkl_split_rows", split_rows)
    for _ in range(2):
        m.insert_pointcloud(pts, origin, -1.0, 0.3, -1.0)
        o.insert_pointcloud(pts, origin, -1.0, 0.3, -1.0)
        _same(m, o, "nan beam")
    assert 0 < int(m.leaves()["classified"].sum()) < m.leaves()["A"].size


def test_against_the_likely_reference_build(built):
    """VERDICT r03 item 2c — the BGK guard of tests/test_bgk_gpu.py for BGKLOctoMap: the HIP path against the restatement
    in its 'likely reference build' modes (oracle.set_modes(1, 1): Eigen 3.3.7 SSE packet sin / cos, pcl::VoxelGrid's
    unstable sort).  Measured (tools/check/likely_ref.py): identical leaf structure, states and
    `classified`, max |dp| 2.6e-6 on three fused sim_structured scans and 1.0e-5 on the synthetic 50 k-ray scan.  The
    bounds guard those numbers; they are not a claim of bit identity with any build."""
    import la3dm_amd
    from oracle import oracle as O
    cases = [("sim_structured x3", [la3dm_amd.load_pcd(pcd_path("sim_structured", i)) for i in (1, 2, 3)], 8.0, False, 5e-6, 1.0),
             ("synthetic 50 k rays", [la3dm_amd.synthetic_scan(50000)], -1.0, True, 2e-5, 0.9999)]
    for tag, scans, max_range, omp, bound, share in cases:
        params = dict(la3dm_amd.L_YAML)
        m = la3dm_amd.BGKLOctoMap(**params, device=0)
        O.set_modes(1, 1, omp=omp)
        try:
            o = O.OracleLMap(**params, omp=omp)
            for xyz, origin in scans:
                m.insert_pointcloud(xyz, origin, 0.1, 0.3, max_range)
                o.insert_pointcloud(xyz, origin, 0.1, 0.3, max_range)
        finally:
            O.set_modes(0, 0, omp=omp)
        a, b = m.leaves(), o.leaves()
        assert a["block_key"].size == b["block_key"].size, tag
        for k in ("block_key", "node_key", "state", "classified"):
            assert (a[k] == b[k]).all(), (tag, k, int((a[k] != b[k]).sum()))
        pa = a["A"].astype(np.float64) / (a["A"].astype(np.float64) + a["B"])
        pb = b["A"].astype(np.float64) / (b["A"].astype(np.float64) + b["B"])
        d = np.abs(pa - pb)
        assert d.max() <= bound, (tag, float(d.max()))
        assert (d <= 1e-5).mean() >= share, (tag, float((d <= 1e-5).mean()))
