"""occupancy sensitivity of the BGK predict kernel: the same packed configs[1] scan with the launch's dynamic LDS padded so
that a CU holds 8 / 7 / 6 / 5 / 4 / 3 waves per SIMD.  gpurun -- python tools/check/occ_sweep.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import la3dm_amd
from r_check import run

if __name__ == "__main__":
    xyz, origin = la3dm_amd.synthetic_scan(200000, seed=1234)
    m = la3dm_amd.BGKOctoMap(**la3dm_amd.BGK_YAML, device=0).set_device_resident(False)
    assert m.prepare(xyz, origin, 0.1, 0.5, -1.0)
    pk = m.packed()
    for mode in (1, 0):
        for pad in (0, 640, 1536, 3072, 5120, 8192):
            _, t = run(m, pk, mode, 10, opts=(("lds_pad", pad),))
            print(f"bgk_sum {mode} lds_pad {pad:5d} (LDS/wave {5120 + pad}, waves/CU {163840 // (5120 + pad)}): kernel ms median {np.median(t):.4f} min {t.min():.4f}")
    m.set_option("lds_pad", 0)
