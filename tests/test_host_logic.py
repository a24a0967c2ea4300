"""CPU tests of the PRODUCT's host side (not gpu): the C-ABI library loads and exports every
declared symbol, refuses to run without a device, and the host-side BGKOctoMap (block hashing,
front end, partition, packing, commit, prune) agrees with the oracle.  The device step is
emulated here with the oracle's per-block predict (the oracle is only the checker)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, pcd_path

YAML = dict(resolution=0.1, block_depth=3, sf2=1.0, ell=0.2, free_thresh=0.3, occupied_thresh=0.7, var_thresh=100.0,
            prior_A=0.001, prior_B=0.001)


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(la3dm_\w+)\s*\(", txt)))


def test_abi_exports_every_declared_symbol(built):
    from la3dm_amd import _lib
    h = C.CDLL(_lib.HIP_SO, mode=C.RTLD_GLOBAL)
    names = _declared("la3dm_hip.h")
    assert len(names) >= 10
    for n in names:
        assert hasattr(h, n), n
    assert sorted(names) == sorted(_lib.HIP_SYMBOLS)
    m = C.CDLL(_lib.MAP_SO)
    mnames = [n for n in _declared("la3dm_map.h")]
    for n in mnames:
        assert hasattr(m, n), n
    assert sorted(mnames) == sorted(_lib.MAP_SYMBOLS)


def test_no_device_no_fallback(built):
    """without a GPU the product refuses to run instead of computing on the CPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import la3dm_amd
    from la3dm_amd import _lib
    with pytest.raises(RuntimeError, match="no HIP device"):
        la3dm_amd.BGKOctoMap(**YAML, device=0)
    m = la3dm_amd.BGKOctoMap(**YAML, device=-1)      # bookkeeping-only map
    with pytest.raises(RuntimeError, match="no device context"):
        m.insert_pointcloud(np.zeros((4, 3), np.float32), [0, 0, 0], 0.1)
    p = _lib.Params()
    out = C.c_void_p()
    assert _lib.hip().la3dm_create(C.byref(p), C.byref(out)) == -1    # LA3DM_ERR_ARG (no LUT)
    assert b"bad params" in _lib.hip().la3dm_last_error(None)


@pytest.mark.parametrize("depth", [3, 4, 5])
def test_host_hash_lut_against_reference_kat(built, depth):
    import la3dm_amd
    kat = np.load(os.path.join(GOLDEN, "ref_kat.npz"))
    m = la3dm_amd.BGKOctoMap(**dict(YAML, block_depth=depth), device=-1)
    assert np.float32(m.get_block_size()) == kat[f"d{depth}_block_size"]
    for p, k, c, e in zip(kat[f"d{depth}_hash_pts"], kat[f"d{depth}_hash_keys"], kat[f"d{depth}_hash_centres"],
                          kat[f"d{depth}_eblocks"]):
        assert m.block_to_hash_key(*map(float, p)) == k
        assert (m.hash_key_to_block(int(k)) == c).all()
        assert (m.get_extended_block(int(k)) == e).all()
    assert (m.lut() == kat[f"d{depth}_lut"]).all()


@pytest.mark.parametrize("depth", [3, 4, 5])
def test_host_lv_hash_lut_against_reference_lv_kat(built, depth):
    """the host-side BGKLVOctoMap's block hashing / ExtendedBlock / voxel LUT against the reference's compiled BGK-LV
    sources (tests/golden/ref_kat_lv.npz: bgklvblock.cpp, depth 5 at 0.05 m is configs[3])"""
    import la3dm_amd
    kat = np.load(os.path.join(GOLDEN, "ref_kat_lv.npz"))
    tag = f"d{depth}"
    m = la3dm_amd.BGKLVOctoMap(**dict(la3dm_amd.LV_YAML, resolution=float(kat[f"{tag}_resolution"]), block_depth=depth), device=-1)
    assert np.float32(m.get_block_size()) == kat[f"{tag}_block_size"]
    for p, k, c, e in zip(kat[f"{tag}_hash_pts"], kat[f"{tag}_hash_keys"], kat[f"{tag}_hash_centres"], kat[f"{tag}_eblocks"]):
        assert m.block_to_hash_key(*map(float, p)) == k
        assert (m.hash_key_to_block(int(k)) == c).all()
        assert (m.get_extended_block(int(k)) == e).all()
    assert (m.lut() == kat[f"{tag}_lut"]).all()


@pytest.mark.parametrize("cls", ["BGKOctoMap", "GPOctoMap", "BGKLOctoMap", "BGKLVOctoMap"])
def test_set_resolution_and_set_block_depth(built, cls):
    """BGKOctoMap::set_resolution / set_block_depth (reference src/bgkoctomap/bgkoctomap.cpp:66-80; the same pair in
    gpoctomap.cpp:55-69, bgkloctomap.cpp:67-81, bgklvoctomap.cpp:73-87): block size and voxel LUT after the call are the
    reference's own for that depth (tests/golden/ref_kat.npz, captured from its compiled sources), hashing follows the
    new block size; on a map that holds blocks the call is an error instead of the reference's silent corruption."""
    import la3dm_amd
    kat = np.load(os.path.join(GOLDEN, "ref_kat.npz"))
    yaml = {"BGKOctoMap": la3dm_amd.BGK_YAML, "GPOctoMap": la3dm_amd.GP_YAML, "BGKLOctoMap": la3dm_amd.L_YAML,
            "BGKLVOctoMap": la3dm_amd.LV_YAML}[cls]
    m = getattr(la3dm_amd, cls)(**dict(yaml, resolution=0.1, block_depth=3), device=-1)
    for depth in (4, 5, 3):
        m.set_block_depth(depth)
        assert m.get_block_depth() == depth and m.get_resolution() == np.float32(0.1)
        assert np.float32(m.get_block_size()) == kat[f"d{depth}_block_size"]
        assert (m.lut() == kat[f"d{depth}_lut"]).all()
        for p, k, c in zip(kat[f"d{depth}_hash_pts"], kat[f"d{depth}_hash_keys"], kat[f"d{depth}_hash_centres"]):
            assert m.block_to_hash_key(*map(float, p)) == k and (m.hash_key_to_block(int(k)) == c).all()
    m.set_resolution(0.25)
    assert m.get_resolution() == np.float32(0.25) and m.get_block_size() == np.float32(4 * 0.25)
    fresh = getattr(la3dm_amd, cls)(**dict(yaml, resolution=0.25, block_depth=3), device=-1)
    assert (m.lut() == fresh.lut()).all()
    for bad in (0.0, -1.0, float("nan")):
        with pytest.raises(RuntimeError):
            m.set_resolution(bad)
    for bad in (0, 7):
        with pytest.raises(RuntimeError):
            m.set_block_depth(bad)
    if cls == "BGKOctoMap":      # a map with blocks refuses
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
        assert m.prepare(xyz, origin, 0.25, 0.5, 8.0)
        m.commit()
        assert m.block_count() > 0
        with pytest.raises(RuntimeError):
            m.set_resolution(0.1)
        with pytest.raises(RuntimeError):
            m.set_block_depth(4)


def _emulate_device(pk, params):
    """what la3dm_bgk_scan_* computes, done with the oracle's predict + node update"""
    from oracle import oracle as O
    o = O.OracleMap(**params)
    lut = np.concatenate(o.lut())
    base = [(8 ** d - 1) // 7 for d in range(8)]
    a, b, s = C.c_float(), C.c_float(), C.c_uint8()
    for t in range(pk.n_test_blk):
        l0, l1 = int(pk.leaf_off[t]), int(pk.leaf_off[t + 1])
        keys = pk.leaf_key[l0:l1]
        xs = lut[[base[k >> 16] + (k & 0xFFFF) for k in keys]] + pk.blk_center[t]
        for nb in pk.nbr[t]:
            if nb < 0:
                continue
            p0, p1 = int(pk.train_off[nb]), int(pk.train_off[nb + 1])
            yb, kb = O.bgk_predict(params["sf2"], params["ell"], xs, pk.train_xyzy[p0:p1, :3], pk.train_xyzy[p0:p1, 3])
            for j in np.nonzero(kb > 0)[0]:
                a.value, b.value, s.value = pk.alpha[l0 + j], pk.beta[l0 + j], pk.state[l0 + j] & 3
                o.L.orc_node_update(o.h, C.byref(a), C.byref(b), C.byref(s), float(yb[j]), float(kb[j]))
                pk.alpha[l0 + j], pk.beta[l0 + j], pk.state[l0 + j] = a.value, b.value, s.value | 0x80


@pytest.mark.parametrize("depth", [3, 4])
def test_prepare_pack_commit_against_oracle(built, depth):
    """host front end + partition + pack + commit + prune == oracle, over three fused scans"""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(YAML, block_depth=depth)
    m = la3dm_amd.BGKOctoMap(**params, device=-1)
    o = O.OracleMap(**params)
    for i in (1, 2, 3):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        assert m.prepare(xyz, origin, 0.1, 0.5, 8.0)
        assert (m.training_data() == O.get_training_data(xyz, origin, 0.1, 0.5, 8.0)).all()
        pk = m.packed()
        # structural checks of the packed scan
        assert pk.leaf_off[0] == 0 and pk.leaf_off[-1] == pk.n_leaf and (np.diff(pk.leaf_off.astype(np.int64)) > 0).all()
        assert (pk.nbr < pk.n_train_blk).all() and (pk.nbr >= -1).all()
        assert pk.train_off[-1] == pk.n_train_pts and (np.diff(pk.train_off.astype(np.int64)) > 0).all()
        w = [sum(int(pk.train_off[n + 1] - pk.train_off[n]) for n in row if n >= 0) for row in pk.nbr]
        assert w == sorted(w, reverse=True)          # heaviest test blocks first
        _emulate_device(pk, params)
        m.commit()
        o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        st, so = m.stats(), o.stats()
        for k in ("n_hits", "n_frees", "n_bbox_blocks", "n_train_blocks", "n_test_blocks", "voxel_updates",
                  "pair_evals", "train_reads"):
            assert st[k] == so[k], k
        a, b = m.leaves(), o.leaves()
        for k in ("block_key", "node_key", "loc", "size", "A", "B", "state", "classified"):
            assert a[k].shape == b[k].shape and (a[k] == b[k]).all(), (i, k)
    if depth == 3:
        assert (a["node_key"] >> 16).min() < 2        # pruning produced coarse leaves


def test_search_bbox_and_iteration(built):
    import la3dm_amd
    from oracle import oracle as O
    m = la3dm_amd.BGKOctoMap(**YAML, device=-1)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    m.prepare(xyz, origin, 0.1, 0.5, 8.0)
    _emulate_device(m.packed(), YAML)
    m.commit()
    lv = m.leaves()
    i = int(np.argmax(lv["A"]))
    e, a, b, s = m.search(*map(float, lv["loc"][i]))
    assert e and np.float32(a) == lv["A"][i] and np.float32(b) == lv["B"][i] and s == lv["state"][i]
    assert m.search(500.0, 500.0, 500.0)[0] is False
    lo, hi = m.get_bbox()
    assert (lo <= lv["loc"].min(0)).all() and (hi >= lv["loc"].max(0)).all()
    assert m.block_count() == len(set(lv["block_key"].tolist()))


def test_block_grid_against_reference_kat(built):
    """Block::get_index / get_node / get_point (the RayCaster's primitives): host code and oracle vs answers
    captured from the reference's own compiled sources (tests/golden/ref_kat_grid.npz).

    Pinned at block_depth 4 only: the reference initialises the static Block::cell_num once from its default
    statics (0.8 / 0.1 = 8 cells, src/bgkoctomap/bgkblock.cpp:103-105) and its constructor never updates it
    (src/bgkoctomap/bgkoctomap.cpp:31-56), so for any other depth its get_index / search(point) address cells
    that do not exist (the depth-3 fixture shows indices up to 7 in a 4-cell block).  This build uses
    cell_num = 2^(block_depth-1), which is what the reference computes for its default depth."""
    import la3dm_amd
    depth = 4
    quirk = np.load(os.path.join(GOLDEN, "ref_kat_grid.npz"))["d3_c0_idx"]
    assert quirk.max() > 3
    from la3dm_amd import _lib
    from oracle import oracle as O
    kat = np.load(os.path.join(GOLDEN, "ref_kat_grid.npz"))
    params = dict(YAML, block_depth=depth)
    m = la3dm_amd.BGKOctoMap(**params, device=-1)
    o = O.OracleMap(**params)
    M = _lib.maplib()
    idx, key, pt = np.zeros(3, np.int32), C.c_int32(), np.zeros(3, np.float32)
    for case in (0, 1):
        tag = f"d{depth}_c{case}"
        c = np.ascontiguousarray(kat[f"{tag}_center"])
        for p, ri, rk, rp in zip(kat[f"{tag}_pts"], kat[f"{tag}_idx"], kat[f"{tag}_key"], kat[f"{tag}_point"]):
            p = np.ascontiguousarray(p)
            M.la3dm_map_block_grid(m._h, c, p, idx, C.byref(key), pt)
            assert (idx == ri).all() and key.value == rk and (pt == rp).all(), ("host", p)
            o.L.orc_block_grid(o.h, c, p, idx, C.byref(key), pt)
            assert (idx == ri).all() and key.value == rk and (pt == rp).all(), ("oracle", p)


def test_raycaster_against_oracle(built):
    """RayCaster walks: host class vs oracle restatement on the same block set (voxel centres, keys, validity),
    plus a hand-checkable axis-aligned walk."""
    import la3dm_amd
    from oracle import oracle as O
    m = la3dm_amd.BGKOctoMap(**YAML, device=-1)
    o = O.OracleMap(**YAML)
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    m.prepare(xyz, origin, 0.1, 0.5, 8.0)           # creates the blocks (nodes stay at the prior)
    o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    rng = np.random.default_rng(11)
    lv = o.leaves()
    starts = lv["loc"][rng.integers(0, lv["loc"].shape[0], 60)] + rng.uniform(-0.04, 0.04, (60, 3)).astype(np.float32)
    ends = starts + rng.uniform(-3, 3, (60, 3)).astype(np.float32)
    ends[:10, 1:] = starts[:10, 1:]                 # axis-aligned
    ends[10:20, 2] = starts[10:20, 2]               # planar
    ends[20:25] = starts[20:25] + np.float32(0.7) * np.sign(rng.uniform(-1, 1, (5, 3))).astype(np.float32)  # diagonals
    nsteps = 0
    for s3, e3 in zip(starts, ends):
        a, b = m.raycast(s3, e3), o.raycast(s3, e3)
        assert a["p"].shape == b["p"].shape and a["p"].shape[0] >= 1
        for k in ("p", "block_key", "node_key", "valid"):
            assert (a[k] == b[k]).all(), k
        nsteps += a["p"].shape[0]
    assert nsteps > 500
    # outside the map nothing starts
    assert m.raycast([500, 500, 500], [501, 500, 500])["p"].shape[0] == 0
    # axis-aligned: consecutive voxel centres one resolution apart, one row per voxel
    s3 = lv["loc"][int(np.argmin(np.abs(lv["loc"] - lv["loc"].mean(0)).sum(1)))]
    w = m.raycast(s3, s3 + np.array([1.25, 0, 0], np.float32))
    assert w["p"].shape[0] == 1 + abs(int((s3[0] + 1.25) / 0.1) - int(s3[0] / 0.1))
    v = w["valid"].astype(bool)
    d = np.diff(w["p"][:, 0])[v[1:] & v[:-1]]
    assert np.allclose(d, 0.1, atol=1e-5) and (w["p"][v][:, 1:] == w["p"][v][0, 1:]).all()


def _sorted_cells(e):
    rows = np.concatenate([e["cells"], e["rgba"], e["level"][:, None].astype(np.float32)], axis=1)
    return rows[np.lexsort(rows.T[::-1])]


def test_height_map_color_known_answers(built):
    """heightMapColor (reference markerarray_pub.h:21-76, s = v = 1) at hand-derived points of the HSV wheel"""
    from oracle import oracle as O
    L = O.lib()
    kat = {0.0: (1, 0, 0), 1 / 6: (1, 1, 0), 0.25: (0.5, 1, 0), 0.5: (0, 1, 1), 0.8: (0.8, 0, 1), 1.0: (1, 0, 0),
           2 / 3: (0, 0, 1), 1.25: (0.5, 1, 0)}
    for h, rgb in kat.items():
        c = np.zeros(4, np.float32)
        L.orc_height_map_color(h, c)
        assert np.allclose(c[:3], rgb, atol=2e-7) and c[3] == 1.0, (h, c)


@pytest.mark.parametrize("depth", [3, 4])
def test_export_cells_against_oracle(built, depth):
    """cube lists of the map (the static node's publish loop + MarkerArrayPub): host class vs oracle restatement, both
    states, original and expanded sizes, bbox-derived and explicit height range"""
    import la3dm_amd
    from oracle import oracle as O
    params = dict(YAML, block_depth=depth)
    m = la3dm_amd.BGKOctoMap(**params, device=-1)
    o = O.OracleMap(**params)
    for i in (1, 2):
        xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", i))
        assert m.prepare(xyz, origin, 0.1, 0.5, 8.0)
        _emulate_device(m.packed(), params)
        m.commit()
        o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    lo, hi = m.get_bbox()
    olo, ohi = o.get_bbox()
    assert (lo == olo).all() and (hi == ohi).all()
    n_total = 0
    for state in ("occupied", "free"):
        for original in (True, False):
            for zr in ((0.0, 0.0), (-0.5, 2.0), (3.0, 1.0)):
                a = m.export_cells(state, original, *zr)
                b = o.export_cells(state, original, *zr)
                assert a["cells"].shape == b["cells"].shape and a["cells"].shape[0] > 0, (state, original, zr)
                assert (_sorted_cells(a) == _sorted_cells(b)).all(), (state, original, zr)
                n_total += a["cells"].shape[0]
                if not original:
                    assert (a["level"] == 0).all() and (a["cells"][:, 3] == np.float32(0.1)).all()
                if state == "occupied" and zr == (3.0, 1.0):
                    assert (a["rgba"] == np.array([0, 0, 1, 1], np.float32)).all()   # uncoloured: the marker default
    lv = m.leaves()
    e = m.export_cells("occupied", True)
    assert e["cells"].shape[0] == int((lv["state"] == 1).sum())
    assert set(np.unique(e["level"]).tolist()) <= set(range(depth))
    assert e["level"].max() >= 1                                                     # pruned (coarse) occupied leaves
    assert m.export_cells("occupied", False)["cells"].shape[0] > e["cells"].shape[0]
    empty = la3dm_amd.BGKOctoMap(**params, device=-1).export_cells("free", True)
    assert empty["cells"].shape == (0, 4)


def test_cpp_example_builds_and_fails_loudly_without_gpu(built):
    """examples/static_map.cpp (the reference's static node loop against the C++ class) is built by build(); without a
    HIP device it must stop with an error — there is no CPU inference path to fall back to"""
    import subprocess
    exe = os.path.join(ROOT, "examples", "static_map")
    assert os.path.exists(exe)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_bgk_gpu.py::test_cpp_example")
    r = subprocess.run([exe, os.path.join(GOLDEN, "data", "sim_structured"), "sim_structured", "1"], capture_output=True,
                       text=True, timeout=120)
    assert r.returncode == 1 and "no HIP device" in r.stderr, (r.returncode, r.stderr)


def test_synthetic_scan_generator(built):
    import la3dm_amd
    xyz, origin = la3dm_amd.synthetic_scan(5000)
    assert xyz.shape == (5000, 3) and xyz.dtype == np.float32 and (origin == [0, 0, 1]).all()
    assert np.abs(xyz[:, :2]).max() < 10.2 and -0.1 < xyz[:, 2].min() and xyz[:, 2].max() < 5.1
    x2, _ = la3dm_amd.synthetic_scan(5000)
    assert (x2 == xyz).all()


def test_two_live_maps_with_different_parameters_keep_their_own(built):
    """the node / block parameters are process-global statics as in the reference (bgkoctomap.cpp:31-56);
    every map re-installs its own set at each public entry point, so a second live map with another depth, resolution
    or variant does not change the first one's hashing, leaf layout or classification."""
    import la3dm_amd
    a = la3dm_amd.BGKOctoMap(**dict(la3dm_amd.BGK_YAML, resolution=0.1, block_depth=3), device=-1)
    ka = a.block_to_hash_key(0.39, -0.41, 7.45)
    lut_a = a.lut().copy()
    b = la3dm_amd.BGKOctoMap(**dict(la3dm_amd.BGK_YAML, resolution=0.25, block_depth=5), device=-1)   # now the bound one
    kb = b.block_to_hash_key(0.39, -0.41, 7.45)
    assert a.block_to_hash_key(0.39, -0.41, 7.45) == ka == 0x800017ffff80013          # SURVEY 9.3, depth 3 / 0.1 m
    assert b.block_to_hash_key(0.39, -0.41, 7.45) == kb != ka
    assert abs(a.get_block_size() - 0.4) < 1e-6 and abs(b.get_block_size() - 4.0) < 1e-6
    assert (a.lut() == lut_a).all() and a.lut().shape[0] == 73 and b.lut().shape[0] == (8 ** 5 - 1) // 7
    # interleaved host-orchestrated work on both maps
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1, 1, (200, 3)).astype(np.float32)
    assert a.prepare(pts, [0, 0, 0.5], 0.1, 0.5, -1.0)
    na = a.packed().n_leaf
    assert b.prepare(pts, [0, 0, 0.5], 0.25, 0.5, -1.0)
    assert a.packed().n_leaf == na
    ca, cb = a.hash_key_to_block(ka), b.hash_key_to_block(kb)
    assert abs(ca[2] - 7.6) < 1e-5 and abs(cb[2] - 8.0) < 1e-5


def test_pcl_overload_compiles_and_runs_with_a_pcl_shaped_cloud(built, tmp_path):
    """the PCL overload of insert_pointcloud (host/bgkoctomap.h, LA3DM_WITH_PCL — what the reference's nodes
    call, include/bgkoctomap/bgkoctomap.h:82-84) was never compiled because PCL is not in the image.  A mock with
    pcl::PointCloud<pcl::PointXYZ>'s shape (16-byte points in `.points`, size(), empty()) instantiates it here and drives
    it through a bookkeeping-only map: the call must reach the map's own entry point with stride 4 (and fail there, for
    lack of a device, with the library's message rather than anything about the cloud)."""
    import subprocess
    from conftest import ROOT
    src = tmp_path / "pcl_mock.cpp"
    src.write_text(r'''
#define LA3DM_WITH_PCL 1
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>
#include "la3dm_amd/csrc/host/bgkoctomap.h"
namespace pcl {
struct alignas(16) PointXYZ { float x, y, z, pad; };
template <class P> struct PointCloud {
    std::vector<P> points;
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
};
}
int main() {
    static_assert(sizeof(pcl::PointXYZ) == 16, "PCL's PointXYZ is 16 bytes");
    pcl::PointCloud<pcl::PointXYZ> cloud;
    for (int i = 0; i < 100; ++i) cloud.points.push_back({1.0f + 0.01f * i, 0.5f, 0.25f, 0.0f});
    la3dm::BGKOctoMap map(0.1f, 3, 1.0f, 0.2f, 0.3f, 0.7f, 100.0f, 0.001f, 0.001f, /*device=*/-1);
    try {
        map.insert_pointcloud(cloud, la3dm::point3f(0, 0, 0), 0.1f, 0.5f, -1.0f);
    } catch (const std::exception &e) {
        std::printf("threw: %s\n", e.what());
        return std::strstr(e.what(), "no device context") ? 0 : 2;
    }
    return 3;
}
''')
    exe = tmp_path / "pcl_mock"
    csrc = os.path.join(ROOT, "la3dm_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", ROOT, str(src), "-o", str(exe), "-L", csrc, "-lla3dm_map", "-lla3dm_hip",
                        f"-Wl,-rpath,{csrc}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


def test_node_stream_operators(built, tmp_path):
    """VERDICT r03 (missing #6): the reference node's stream operators (src/bgkoctomap/bgkoctree_node.cpp:46-62).  The
    binary pair writes m_A, m_B as 8 raw bytes and reads them back through the (A, B) constructor (which adds the priors
    again, as the reference's does); the text form is "(m_A m_B prob)"."""
    import subprocess
    from conftest import ROOT
    src = tmp_path / "node_io.cpp"
    out = tmp_path / "node.bin"
    src.write_text(r'''
#include <cstdio>
#include <fstream>
#include <sstream>
#include "la3dm_amd/csrc/host/bgkoctomap.h"
int main(int argc, char **argv) {
    la3dm::BGKOctoMap map(0.1f, 3, 1.0f, 0.2f, 0.3f, 0.7f, 100.0f, 0.001f, 0.001f, /*device=*/-1);   // installs the priors
    la3dm::OcTreeNode n(2.0f, 0.5f);                          // m_A = 0.001 + 2, m_B = 0.001 + 0.5
    { std::ofstream os(argv[1], std::ios::binary); os << n; }
    la3dm::OcTreeNode r;
    { std::ifstream is(argv[1], std::ios::binary); is >> r; }
    std::ostringstream a, b;
    a << n;
    b << r;
    std::printf("%s\n%s\n%d\n", a.str().c_str(), b.str().c_str(), (int)r.get_state());
    return 0;
}
''')
    exe = tmp_path / "node_io"
    csrc = os.path.join(ROOT, "la3dm_amd", "csrc")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I", ROOT, str(src), "-o", str(exe), "-L", csrc, "-lla3dm_map", "-lla3dm_hip",
                        f"-Wl,-rpath,{csrc}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    raw = np.fromfile(out, np.float32)
    A, B = np.float32(0.001) + np.float32(2.0), np.float32(0.001) + np.float32(0.5)
    assert raw.size == 2 and raw[0] == A and raw[1] == B                       # 8 raw bytes: m_A, m_B
    l1, l2, st = r.stdout.strip().splitlines()
    assert l1.startswith("(") and l1.endswith(")") and len(l1.strip("()").split()) == 3
    a1, b1, p1 = (float(v) for v in l1.strip("()").split())
    assert abs(a1 - float(A)) < 1e-4 and abs(b1 - float(B)) < 1e-4 and abs(p1 - float(A) / (float(A) + float(B))) < 1e-5
    a2, b2, _ = (float(v) for v in l2.strip("()").split())
    assert abs(a2 - (float(A) + 0.001)) < 1e-4 and abs(b2 - (float(B) + 0.001)) < 1e-4   # read back through OcTreeNode(A, B)
    assert int(st) == 1                                                          # p = 0.8 > 0.7: OCCUPIED
