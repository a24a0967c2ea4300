"""CPU tests (not gpu) for the GP and LV variants: oracle numerics against float64, host front ends
and bookkeeping against the oracle (bookkeeping-only maps: no device, no compute)."""
import ctypes as C

import numpy as np
import pytest

from conftest import pcd_path


def test_gp_regressor_against_float64(built):
    from oracle import oracle as O
    g = O.OracleGPMap(**O.GP_YAML)
    rng = np.random.default_rng(2)
    for n in (1, 7, 40, 79):
        x = rng.uniform(-0.2, 0.2, (n, 3)).astype(np.float32)
        y = np.where(rng.uniform(size=n) < 0.4, 1, -1).astype(np.float32)
        xs = rng.uniform(-0.25, 0.25, (64, 3)).astype(np.float32)
        a, L, m, v = g.train_predict(x, y, xs)
        s = float(np.float32(1.73205 / 1.0))
        d = np.linalg.norm(x[:, None].astype(np.float64) - x[None], axis=2) * s
        K = (1 + d) * np.exp(-d) + 0.01 * np.eye(n)
        dk = np.linalg.norm(x[:, None].astype(np.float64) - xs[None], axis=2) * s
        ks = (1 + dk) * np.exp(-dk)
        np.testing.assert_allclose(L @ L.T, K, atol=5e-6)                       # Cholesky factor
        assert np.allclose(np.triu(L, 1), 0)
        m64 = ks.T @ np.linalg.solve(K, y.astype(np.float64))
        v64 = 1 - np.einsum("ij,ij->j", ks, np.linalg.solve(K, ks))
        assert np.abs(m - m64).max() < 2e-3 * max(1.0, np.abs(m64).max())        # fp32 on an ill-conditioned K
        assert np.abs(v - v64).max() < 5e-5
    # BCM node update (src/gpoctomap/gpoctree_node.cpp:36-49)
    mi, iv, st = C.c_float(0.0), C.c_float(0.001), C.c_uint8(2)
    g.L.orc_gp_node_update(g.h, C.byref(mi), C.byref(iv), C.byref(st), 0.9, 0.01)
    assert abs(iv.value - (0.001 + 100.0 - 1.0)) < 1e-4 and abs(mi.value - 90.0) < 1e-4 and st.value == 1
    g.L.orc_gp_node_update(g.h, C.byref(mi), C.byref(iv), C.byref(st), -0.9, 0.005)
    assert st.value == 0 and abs(g.L.orc_gp_node_prob(g.h, mi.value) - 1 / (1 + np.exp(0.1 * 90.0))) < 1e-6


def test_gp_eigen_order_mode_against_float64(built):
    """oracle.set_gp_mode(1): the GP restated in Eigen 3.3.7's order of operations (no FMA, SSE packet sums, blocked
    LLT, panelled triangular solves, packet exp) is a valid fp32 evaluation of the same regressor — as close to the
    float64 evaluation as the FMA-chain mode, for block sizes on both sides of the blocking threshold (32) and of a
    panel (8) — and it is a DIFFERENT rounding: the two modes do not agree bit for bit."""
    from oracle import oracle as O
    g = O.OracleGPMap(**O.GP_YAML)
    rng = np.random.default_rng(5)
    differs = 0
    for n in (1, 7, 8, 9, 31, 32, 40, 79, 200):
        x = rng.uniform(-0.4, 0.4, (n, 3)).astype(np.float32)
        y = np.where(rng.uniform(size=n) < 0.4, 1, -1).astype(np.float32)
        xs = rng.uniform(-0.45, 0.45, (64, 3)).astype(np.float32)
        a0, L0, m0, v0 = g.train_predict(x, y, xs)
        O.set_gp_mode(1)
        try:
            a1, L1, m1, v1 = g.train_predict(x, y, xs)
        finally:
            O.set_gp_mode(0)
        s = float(np.float32(1.73205 / 1.0))
        d = np.linalg.norm(x[:, None].astype(np.float64) - x[None], axis=2) * s
        K = (1 + d) * np.exp(-d) + 0.01 * np.eye(n)
        dk = np.linalg.norm(x[:, None].astype(np.float64) - xs[None], axis=2) * s
        ks = (1 + dk) * np.exp(-dk)
        np.testing.assert_allclose(L1 @ L1.T, K, atol=1e-5)
        assert np.allclose(np.triu(L1, 1), 0)
        m64 = ks.T @ np.linalg.solve(K, y.astype(np.float64))
        v64 = 1 - np.einsum("ij,ij->j", ks, np.linalg.solve(K, ks))
        tol_m = 4e-3 * max(1.0, np.abs(m64).max())
        assert np.abs(m1 - m64).max() < tol_m and np.abs(m0 - m64).max() < tol_m, n
        assert np.abs(v1 - v64).max() < 1e-4 and np.abs(v0 - v64).max() < 1e-4, n
        differs += int((a0 != a1).any() or (m0 != m1).any())
    assert differs >= 5


def test_gp_host_front_end_and_hints(built):
    import la3dm_amd
    from oracle import oracle as O
    m = la3dm_amd.GPOctoMap(**la3dm_amd.GP_YAML, device=-1)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    assert m.prepare(xyz, origin, 0.1, 0.1, 8.0)
    t = m.training_data()
    assert set(np.unique(t[:, 3]).tolist()) == {-1.0, 1.0}                       # frees are labelled -1
    o = O.OracleGPMap(**O.GP_YAML)
    o.insert_pointcloud(xyz, origin, 0.1, 0.1, 8.0)
    st, so = m.stats(), o.stats()
    for k in ("n_hits", "n_frees", "n_train_blocks", "n_test_blocks", "voxel_updates", "pair_evals"):
        assert st[k] == so[k], k
    pk = m.packed()
    n = np.diff(pk.train_off.astype(np.int64))
    assert pk.c.train_max_n == n.max() and pk.c.train_sum_n2 == int((n * n).sum())
    lv = m.leaves()
    assert (lv["A"] == 0).all() and np.allclose(lv["B"], 0.001)                  # fresh GP nodes: (0, min_ivar)


def test_lv_oracle_pieces(built):
    from oracle import oracle as O
    L = O.lib()
    f = lambda *v: np.array(v, np.float32)
    # point-to-segment distance: interior projection, both end caps, degenerate segment
    assert abs(L.orc_lv_seg_dist(f(0.5, 1, 0), f(0, 0, 0), f(1, 0, 0)) - 1.0) < 1e-7
    assert abs(L.orc_lv_seg_dist(f(-1, 0, 0), f(0, 0, 0), f(1, 0, 0)) - 1.0) < 1e-7
    assert abs(L.orc_lv_seg_dist(f(3, 0, 0), f(0, 0, 0), f(1, 0, 0)) - 2.0) < 1e-7
    assert abs(L.orc_lv_seg_dist(f(0, 0, 2), f(1, 1, 1), f(1, 1, 1)) - np.sqrt(3)) < 1e-6
    # kernel: r clamped at 1 and NOT clamped at 0 below: k(d >= ell) is the same tiny negative constant
    k1 = L.orc_lv_kernel(0.2, 0.2, 0.1)
    assert k1 == L.orc_lv_kernel(5.0, 0.2, 0.1) and -1e-8 < k1 < 0
    assert L.orc_lv_kernel(0.0, 0.2, 0.1) == np.float32(0.1)
    # LV node: min_W floor, UNCERTAIN state
    m = O.OracleLVMap(**O.LV_YAML)
    assert abs(L.orc_lv_node_prob(m.h, 0.0, 0.0) - 0.5) < 1e-7
    assert L.orc_lv_node_prob(m.h, 5.0, 0.0) == 1.0 and L.orc_lv_node_prob(m.h, 0.0, 5.0) == 0.0
    a, b, s = C.c_float(0.001), C.c_float(0.001), C.c_uint8(2)
    L.orc_lv_node_update(m.h, C.byref(a), C.byref(b), C.byref(s), 0.5, 1.0)      # p = 0.5: variance 0.25 > 0.2
    assert s.value == 3
    L.orc_lv_node_update(m.h, C.byref(a), C.byref(b), C.byref(s), 9.0, 9.0)
    assert s.value == 1


@pytest.mark.parametrize("res,depth", [(0.1, 4), (0.05, 5)])
def test_lv_host_front_end(built, res, depth):
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.LV_YAML, resolution=res, block_depth=depth)
    m = la3dm_amd.BGKLVOctoMap(**params, device=-1)
    o = O.OracleLVMap(**params)
    for i in (1, 5):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_unstructured", i))
        assert m.lv_prepare(xyz, origin, res, 0.1, 8.0)
        s, r = m.lv_training()
        s2, r2 = o.training_data(xyz, origin, res, 0.1, 8.0)
        assert s.shape == s2.shape and (s == s2).all() and (r == r2).all()
        # a ray's samples are contiguous and start with the segment start
        ray = s[:, 3].astype(int)
        first = np.array([np.argmax(ray == k) for k in range(len(r))])
        assert (s[first, :3] == r[:, :3]).all()
        pk = m.lv_packed()
        assert pk.n_samples == len(s) and pk.n_rays == len(r) and pk.n_blk == m.lv_stats()["n_packed_blocks"]


@pytest.mark.parametrize("depth", [3, 4])
def test_bgkl_host_front_end_and_rows(built, depth):
    """BGKLOctoMap host side (bookkeeping-only map): training samples, beam indices and beams == oracle, and the
    per-block segment rows have the oracle's counts (pair_evals / train_reads are counted in rows)"""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.L_YAML, block_depth=depth)
    m = la3dm_amd.BGKLOctoMap(**params, device=-1)
    o = O.OracleLMap(**params)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 2))
    assert m.prepare(xyz, origin, 0.1, 0.3, 8.0)
    xy, rays = m.l_training()
    rxy, rrays = O.l_training_data(xyz, origin, 0.1, 0.3, 8.0)
    assert xy.shape == rxy.shape and rays.shape == rrays.shape
    assert (xy.view(np.uint32) == rxy.view(np.uint32)).all() and (rays.view(np.uint32) == rrays.view(np.uint32)).all()
    o.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
    st, so = m.stats(), o.stats()
    for k in ("n_hits", "n_frees", "n_bbox_blocks", "n_train_blocks", "n_test_blocks", "voxel_updates", "pair_evals",
              "train_reads"):
        assert st[k] == so[k], (k, st[k], so[k])
