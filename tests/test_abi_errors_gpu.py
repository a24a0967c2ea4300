"""Error behaviour of the C-ABI (include/la3dm_hip.h): the reference's API is void / assert-only; this boundary
returns a negative code and leaves the text in la3dm_last_error — it never crashes on a bad argument, never computes
on the CPU instead, and a failed call leaves the map usable."""
import ctypes as C

import numpy as np
import pytest

from conftest import pcd_path

pytestmark = pytest.mark.gpu

OK, ERR_ARG, ERR_NODEVICE = 0, -1, -3


def _err(H, ctx):
    return H.la3dm_last_error(ctx).decode()


def test_scan_entry_points_reject_bad_arguments(built):
    import la3dm_amd
    from la3dm_amd import _lib
    H = _lib.hip()
    m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
    ctx = m.ctx()
    assert H.la3dm_bgk_scan_host(ctx, None, None) == ERR_ARG and "null" in _err(H, ctx)
    assert H.la3dm_bgk_scan_host(None, None, None) == ERR_ARG
    s = _lib.BgkScan()                                   # all zero: no test blocks -> nothing to do
    assert H.la3dm_bgk_scan_host(ctx, C.byref(s), None) == OK
    s.n_test_blk = 3                                     # work announced, arrays missing
    assert H.la3dm_bgk_scan_host(ctx, C.byref(s), None) == ERR_ARG and "null array" in _err(H, ctx)
    assert H.la3dm_bgk_scan_device(ctx, C.byref(s), None, None) == ERR_ARG
    # a BGK context does not run the other variants' entry points
    assert H.la3dm_gp_scan_host(ctx, C.byref(s), None) == ERR_ARG
    s2 = _lib.BgkScan()
    s2.n_test_blk = 1
    dummy = np.zeros(64, np.float32)
    for f in ("nbr", "blk_center", "leaf_off", "leaf_key", "alpha", "beta", "state", "train_off"):
        setattr(s2, f, dummy.ctypes.data)
    assert H.la3dm_gp_scan_host(ctx, C.byref(s2), None) == ERR_ARG and "variant = 1" in _err(H, ctx)
    assert H.la3dm_bgkl_scan_host(ctx, C.byref(s2), None) == ERR_ARG and "variant = 3" in _err(H, ctx)
    # LA3DM_SCAN_ROWS_PREPARED (rows of 12 floats already in HBM) is a flag of la3dm_bgkl_scan_device only: the host forms upload
    # 8 floats per row, so they refuse it instead of letting the kernels read past the upload
    ml = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
    s3 = _lib.BgkScan()
    s3.n_test_blk = 1
    for f in ("nbr", "blk_center", "leaf_off", "leaf_key", "alpha", "beta", "state", "train_off", "train_xyzy"):
        setattr(s3, f, dummy.ctypes.data)
    s3.flags = 0x8
    assert H.la3dm_bgkl_scan_host(ml.ctx(), C.byref(s3), None) == ERR_ARG and "ROWS_PREPARED" in _err(H, ml.ctx())
    assert H.la3dm_set_option(ml.ctx(), b"bgkl_split_rows", -7) == ERR_ARG
    assert H.la3dm_set_option(ml.ctx(), b"bgkl_split_rows", -1) == OK and H.la3dm_set_option(ml.ctx(), b"bgkl_split_rows", 1024) == OK
    assert H.la3dm_set_option(ctx, b"no_such_option", 1) == ERR_ARG and "unknown option" in _err(H, ctx)
    assert H.la3dm_set_option(ctx, None, 1) == ERR_ARG
    # the map still works after the failed calls
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    assert m.block_count() > 100


def test_create_rejects_bad_device_and_params(built):
    import la3dm_amd
    from la3dm_amd import _lib
    H = _lib.hip()
    with pytest.raises(RuntimeError, match="device"):
        la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=99)
    p = _lib.Params()
    out = C.c_void_p()
    assert H.la3dm_create(C.byref(p), C.byref(out)) == ERR_ARG and not out.value
    assert H.la3dm_create(None, C.byref(out)) == ERR_ARG
    H.la3dm_destroy(None)                                 # no-op


def test_device_map_entry_points_reject_bad_arguments(built):
    import la3dm_amd
    from la3dm_amd import _lib
    H = _lib.hip()
    dm = C.c_void_p()
    lv6 = la3dm_amd.BGKLVOctoMap(**dict(la3dm_amd.LV_YAML, block_depth=6), device=0)       # deeper than the pool supports
    assert not lv6.is_device_resident()
    assert H.la3dm_devmap_create(lv6.ctx(), C.byref(dm)) == ERR_ARG and "block_depth" in _err(H, lv6.ctx()) and not dm.value
    assert la3dm_amd.BGKLVOctoMap(**la3dm_amd.LV_YAML, device=0).is_device_resident()        # BGK-LV lives on the pool too
    deep = la3dm_amd.BGKOctoMap(**dict(la3dm_amd.BGK_YAML, block_depth=6), device=0)
    assert not deep.is_device_resident()                  # the class falls back to the host-orchestrated mode
    assert H.la3dm_devmap_create(deep.ctx(), C.byref(dm)) == ERR_ARG and "block_depth" in _err(H, deep.ctx())
    with pytest.raises(RuntimeError):
        deep.set_device_resident(True)
    assert H.la3dm_devmap_create(None, C.byref(dm)) == ERR_ARG
    m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
    assert m.is_device_resident()
    n = C.c_uint64(7)
    # empty pool: nothing to export, no key box
    e = m.export_cells("occupied")
    assert e["cells"].shape == (0, 4)
    lo, hi = m.get_bbox()
    assert (lo == 0).all() and (hi == 0).all()
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    M = m._M
    assert M.la3dm_map_export_cells(m._h, 2, 1, 0.0, 0.0, None, None, None, 0, C.byref(n)) != 0      # UNKNOWN is not exported
    buf = np.zeros(4, np.float32)
    lvl = np.zeros(1, np.int32)
    assert M.la3dm_map_export_cells(m._h, 1, 1, 0.0, 0.0, buf.ctypes.data, buf.ctypes.data, lvl.ctypes.data, 1, C.byref(n)) != 0
    assert b"too small" in M.la3dm_map_last_error()
    # a host-orchestrated map that holds blocks cannot be switched to the device pool afterwards
    h = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0).set_device_resident(False)
    h.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    with pytest.raises(RuntimeError, match="stays host-orchestrated"):
        h.set_device_resident(True)
    # NaN / inf points are dropped by the voxel grid like PCL does; an all-NaN cloud is an empty scan
    bad = np.full((16, 3), np.nan, np.float32)
    before = m.block_count()
    m.insert_pointcloud(bad, origin, 0.1, 0.5, 8.0)
    assert m.block_count() == before


def test_scan_arguments_that_would_hang_the_beam_sampler_are_rejected(built):
    """the beam sampler's `for (d = fr; d < l; d += fr)` (bgkoctomap.cpp:445-457) runs on the GPU; arguments
    that would make it spin are rejected at the ABI (or, when only the data decides, flagged by the kernel) instead of
    hanging the device, and the map stays usable."""
    import la3dm_amd
    for make in (lambda: la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0), lambda: la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)):
        m = make()
        assert m.is_device_resident()
        pts = np.array([[2.0, 0.3, 0.4], [1.5, -1.0, 0.2], [0.2, 2.5, 1.0]], np.float32)
        for fr in (0.0, -0.5, float("nan"), float("inf")):
            with pytest.raises(RuntimeError, match="free_resolution"):
                m.insert_pointcloud(pts, [0, 0, 0.5], 0.1, fr, -1.0)
        with pytest.raises(RuntimeError, match="ds_resolution"):
            m.insert_pointcloud(pts, [0, 0, 0.5], float("nan"), 0.5, -1.0)
        with pytest.raises(RuntimeError, match="origin"):
            m.insert_pointcloud(pts, [0, float("inf"), 0.5], 0.1, 0.5, -1.0)
        # data-dependent: an unfiltered cloud (ds < 0, no range gate) with an infinitely far hit, and a free_resolution
        # far below the fp32 spacing at the beam's range (d += fr stops advancing)
        with pytest.raises(RuntimeError, match="does not terminate"):
            m.insert_pointcloud(pts, [0, 0, 0.5], -1.0, 1e-9, -1.0)
        assert m.block_count() == 0
        far = np.array([[np.inf, 0.0, 1.0]], np.float32)               # an infinite range: must come back, either way
        try:
            m.insert_pointcloud(far, [0, 0, 0.5], -1.0, 0.5, -1.0)
        except RuntimeError as e:
            assert "does not terminate" in str(e)
        before = m.block_count()
        m.insert_pointcloud(pts, [0, 0, 0.5], 0.1, 0.3, -1.0)          # the map is still usable
        assert m.block_count() > before


def test_set_option_validates_its_values(built):
    import la3dm_amd
    m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0)
    for name, bad in (("waves_per_wg", 0), ("waves_per_wg", 3), ("waves_per_wg", 8), ("remap", 3), ("remap", -1), ("fast_trig", 5),
                      ("ablate", 99)):
        with pytest.raises(RuntimeError):
            m.set_option(name, bad)
    for name, ok in (("waves_per_wg", 2), ("waves_per_wg", 1), ("remap", 0), ("remap", 2), ("fast_trig", 0)):
        m.set_option(name, ok)
