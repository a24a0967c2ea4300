"""Device memory is given back: maps of every variant are created, filled and destroyed in a loop; after a warm-up the
free HBM reported by the runtime must not drift (every arena of a context / device pool is released on destroy)."""
import gc

import numpy as np
import pytest

from conftest import pcd_path

pytestmark = pytest.mark.gpu


def test_create_insert_destroy_cycles_do_not_leak(built):
    import torch
    import la3dm_amd
    xyz, origin = la3dm_amd.load_pcd(pcd_path("sim_structured", 1))
    big, big_origin = la3dm_amd.synthetic_scan(12000)       # BGK-L: large enough for the split path's scratch

    def free():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    f0 = None
    for it in range(14):
        for cls, params in ((la3dm_amd.BGKOctoMap, la3dm_amd.BGK_YAML), (la3dm_amd.GPOctoMap, la3dm_amd.GP_YAML),
                            (la3dm_amd.BGKLOctoMap, la3dm_amd.L_YAML), (la3dm_amd.BGKLVOctoMap, la3dm_amd.LV_YAML)):
            m = cls(**params, device=0)
            m.insert_pointcloud(xyz[::4], origin, 0.1, 0.5, 8.0)
            if cls is la3dm_amd.BGKLOctoMap:
                m.set_option("bgkl_split_rows", 500)
                m.insert_pointcloud(big, big_origin, 0.1, 0.3, -1.0)
            if it % 2 and cls is not la3dm_amd.BGKLVOctoMap:
                m.set_device_resident(False)                 # download + destroy the pool, keep the map
            if cls is not la3dm_amd.BGKLVOctoMap:
                m.export_cells("occupied")
            assert m.leaves()["A"].size > 0
            del m
        gc.collect()
        if it == 3:
            f0 = free()
    drift = (f0 - free()) / 2 ** 20
    assert abs(drift) < 1.0, f"free device memory drifted by {drift:.1f} MiB over 10 cycles"
