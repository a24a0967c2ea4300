"""GPU parity tests for GPOctoMap (BASELINE config 3): HIP path vs the CPU oracle.

Tolerance: |dp| <= 1e-5 on the logistic occupancy probability; (m_ivar, ivar) to 1e-5 relative.  The device
uses the oracle's operation order (FMA chains in ascending index order, correctly rounded exp), so
the observed difference is 0; the tolerance only covers a double-rounded exp differing in the last place."""
import numpy as np
import pytest

from conftest import pcd_path

pytestmark = pytest.mark.gpu


def _prob(l, max_ivar, m_ivar):
    return 1.0 / (1.0 + np.exp(-l * m_ivar.astype(np.float64) / max_ivar))


def _compare(m, o, params, tag):
    a, b = m.leaves(), o.leaves()
    assert a["block_key"].size == b["block_key"].size and (a["block_key"] == b["block_key"]).all(), tag
    assert (a["node_key"] == b["node_key"]).all() and (a["loc"] == b["loc"]).all(), tag
    assert (a["classified"] == b["classified"]).all(), tag
    np.testing.assert_allclose(a["A"], b["A"], rtol=1e-5, atol=1e-6, err_msg=tag)   # m_ivar
    np.testing.assert_allclose(a["B"], b["B"], rtol=1e-5, atol=1e-6, err_msg=tag)   # ivar
    pa = _prob(params["l"], 1.0 / params["min_var"], a["A"])
    pb = _prob(params["l"], 1.0 / params["min_var"], b["A"])
    assert np.abs(pa - pb).max() <= 1e-5, tag
    assert (a["state"] == b["state"]).mean() >= 0.9999, tag
    return float((a["A"] == b["A"]).mean()), float((a["B"] == b["B"]).mean())


@pytest.mark.parametrize("depth", [3, 4])
def test_gp_sim_structured(built, depth):
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.GP_YAML, block_depth=depth)
    m = la3dm_amd.GPOctoMap(**params, device=0)
    o = O.OracleGPMap(**params)
    n = 3 if depth == 3 else 1
    for i in range(1, n + 1):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        if depth == 4:
            xyz = xyz[::4]          # keeps the per-block N (and the CPU oracle's N^3) moderate
        m.insert_pointcloud(xyz, origin, 0.1, 0.1, 8.0)
        o.insert_pointcloud(xyz, origin, 0.1, 0.1, 8.0)
        ea, eb = _compare(m, o, params, f"d{depth} scan{i}")
    assert ea > 0.999 and eb > 0.999    # in practice bit-identical


def test_gp_block_kat(built):
    """one block through the whole map path: a single training block, its 7 test blocks"""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.GP_YAML)
    m = la3dm_amd.GPOctoMap(**params, device=0)
    o = O.OracleGPMap(**params)
    rng = np.random.default_rng(11)
    pts = rng.uniform(-0.19, 0.19, (30, 3)).astype(np.float32)
    m.insert_pointcloud(pts, [0.0, 0.0, 3.0], -1.0, 0.35, -1.0)
    o.insert_pointcloud(pts, [0.0, 0.0, 3.0], -1.0, 0.35, -1.0)
    _compare(m, o, params, "kat")


def test_mfma_accumulates_like_an_fma_chain(built):
    """the property the MFMA Cholesky / solve rely on: v_mfma_f32_32x32x2_f32 == fmaf chains over k ascending,
    bit for bit, for values of mixed magnitude and sign"""
    import ctypes as C
    import la3dm_amd
    from la3dm_amd import _lib
    m = la3dm_amd.GPOctoMap(**la3dm_amd.GP_YAML, device=0)
    rng = np.random.default_rng(9)
    for K in (2, 34, 64, 530):
        A = (rng.uniform(-1, 1, (32, K)) * 10.0 ** rng.integers(-4, 3, (32, K))).astype(np.float32)
        B = (rng.uniform(-1, 1, (K, 32)) * 10.0 ** rng.integers(-4, 3, (K, 32))).astype(np.float32)
        bad = C.c_uint32(123)
        assert _lib.hip().la3dm_diag_mfma_chain(m.ctx(), A.ctypes.data, B.ctypes.data, K, C.byref(bad)) == 0
        assert bad.value == 0, (K, bad.value)


def test_gp_large_blocks_on_matrix_cores(built):
    """depth 4, dense scan: training blocks with hundreds of points go through the MFMA Cholesky and the MFMA
    forward substitution; every leaf must still equal the oracle bit for bit"""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(la3dm_amd.GP_YAML, block_depth=4)
    m = la3dm_amd.GPOctoMap(**params, device=0)
    o = O.OracleGPMap(**params, omp=True)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    xyz = xyz[::2]
    m.insert_pointcloud(xyz, origin, 0.1, 0.1, 8.0)
    o.insert_pointcloud(xyz, origin, 0.1, 0.1, 8.0)
    a, b = m.leaves(), o.leaves()
    for k in ("block_key", "node_key", "state", "classified"):
        assert (a[k] == b[k]).all(), k
    for k in ("A", "B"):
        assert (a[k].view(np.uint32) == b[k].view(np.uint32)).all(), (k, float(np.abs(a[k] - b[k]).max()))
