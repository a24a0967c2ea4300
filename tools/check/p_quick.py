"""bgk_predict_fuse_p (tile records, sin / cos table in LDS) against bgk_predict_fuse_t on the same packed scans:
bit identity of alpha / beta / state and kernel times (+ by ablation).   gpurun -- python tools/check/p_quick.py [--big]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import la3dm_amd
from r_check import run

MODES = (0, 1)   # bgk_p: 0 = bgk_predict_fuse_t (default), 1 = bgk_predict_fuse_p


def one(name, xyz, origin, res, depth, reps, fr=0.5, mr=-1.0, ablate=False, strip_full=False):
    m = la3dm_amd.BGKOctoMap(**dict(la3dm_amd.BGK_YAML, resolution=res, block_depth=depth), device=0).set_device_resident(False)
    assert m.prepare(xyz, origin, res, fr, mr)
    pk = m.packed()
    if strip_full:
        pk.c.flags &= ~4   # without LA3DM_SCAN_FULL_BLOCKS: the instances that carry the general path
    ref = None
    for mode in MODES:
        (a, b, s), t = run(m, pk, 1, reps, opts=(("bgk_tables", 1), ("bgk_p", mode)))
        if ref is None:
            ref = (a, b, s)
        line = (f"{name} bgk_p {mode}: {np.median(t):.4f} ms (min {t.min():.4f})  differ alpha {int((a != ref[0]).sum())} "
                f"beta {int((b != ref[1]).sum())} state {int((s != ref[2]).sum())}")
        if ablate:
            ab = []
            for k in (1, 2):
                _, t2 = run(m, pk, 1, 5, opts=(("ablate", k), ("bgk_p", mode)))
                ab.append(float(np.median(t2)))
            m.set_option("ablate", 0)
            line += f"  (no C {ab[0]:.4f}, no B/C {ab[1]:.4f})"
        print(line, flush=True)
    m.set_option("bgk_p", 0)


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    xyz, origin = la3dm_amd.load_pcd(os.path.join(root, "tests/golden/data/sim_structured/sim_structured_1.pcd"))
    if "--only-big" in sys.argv:
        xyz, origin = la3dm_amd.synthetic_scan(1000000, seed=1234)
        one("1M 0.05 d3", xyz, origin, 0.05, 3, 8)
        sys.exit(0)
    one("sim_structured_1 d3", xyz, origin, 0.1, 3, 3, mr=8.0)
    one("sim_structured_1 d4", xyz, origin, 0.1, 4, 3, mr=8.0)
    one("sim_structured_1 d3 general", xyz, origin, 0.1, 3, 3, mr=8.0, strip_full=True)
    xyz, origin = la3dm_amd.synthetic_scan(200000, seed=1234)
    one("200k d3", xyz, origin, 0.1, 3, 20, ablate=True)
    one("200k d3 general", xyz, origin, 0.1, 3, 10, strip_full=True)
    one("200k d4", xyz, origin, 0.1, 4, 5)
    if "--big" in sys.argv:
        xyz, origin = la3dm_amd.synthetic_scan(1000000, seed=1234)
        one("1M 0.05 d3", xyz, origin, 0.05, 3, 5)
