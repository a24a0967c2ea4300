"""GPU parity against the COMMITTED full-scan fixtures (tests/golden/scan_kat.npz, written by
tests/golden/make_golden.py from the CPU restatement): the HIP path must reproduce every leaf of BASELINE configs[0]
(sim_structured scan 1) for each map variant bit for bit, without the live oracle in the loop."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, pcd_path

pytestmark = pytest.mark.gpu

KEYS = ("block_key", "node_key", "A", "B", "state", "classified")


def _check(tag, lv, kat):
    for k in KEYS:
        ref = kat[f"{tag}_{k}"]
        got = lv[k][:ref.shape[0]] if lv[k].shape[0] >= ref.shape[0] else lv[k]
        assert lv[k].shape[0] == ref.shape[0], (tag, k, lv[k].shape, ref.shape)
        if ref.dtype == np.float32:
            assert (got.view(np.uint32) == ref.view(np.uint32)).all(), (tag, k, float(np.abs(got - ref).max()))
        else:
            assert (got.astype(ref.dtype) == ref).all(), (tag, k)


@pytest.fixture(scope="module")
def kat():
    return np.load(os.path.join(GOLDEN, "scan_kat.npz"))


@pytest.mark.parametrize("mode", ["host-orchestrated", "device-resident"])
@pytest.mark.parametrize("depth", [3, 4])
def test_bgk(built, kat, depth, mode):
    import la3dm_amd
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    m = la3dm_amd.BGKOctoMap(**dict(la3dm_amd.BGK_YAML, block_depth=depth), device=0)
    m.set_device_resident(mode == "device-resident")
    assert m.is_device_resident() == (mode == "device-resident")
    m.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
    _check(f"bgk_d{depth}", m.leaves(), kat)


def test_gp(built, kat):
    import la3dm_amd
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    m = la3dm_amd.GPOctoMap(**la3dm_amd.GP_YAML, device=0)
    m.insert_pointcloud(xyz[::3], origin, 0.1, 0.5, 8.0)
    _check("gp_d3", m.leaves(), kat)


def test_bgkl(built, kat):
    import la3dm_amd
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    m = la3dm_amd.BGKLOctoMap(**la3dm_amd.L_YAML, device=0)
    m.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
    _check("bgkl_d3", m.leaves(), kat)


def test_bgklv(built, kat):
    import la3dm_amd
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_unstructured", 1))
    m = la3dm_amd.BGKLVOctoMap(**dict(la3dm_amd.LV_YAML, resolution=0.05), device=0)
    m.insert_pointcloud(xyz, origin, 0.05, 0.1, 8.0)
    _check("lv_d5", m.leaves(), kat)
