"""bgk_predict_fuse_t (distance tables) against bgk_predict_fuse_r on the same packed scans: alpha / beta / state must agree
(both are double sums rounded once: only the order of the double additions differs), and the kernel times.
   gpurun -- python tools/check/t_check.py [--big]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import la3dm_amd
from r_check import run


def compare(name, xyz, origin, res, depth, reps=10, fr=0.5, mr=-1.0, scans=1):
    params = dict(la3dm_amd.BGK_YAML, resolution=res, block_depth=depth)
    m = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
    for s in range(scans - 1):       # earlier scans fused and committed: the packed scan then meets a pruned map
        assert m.prepare(xyz + np.float32(0.013 * (s + 1)), origin, res, fr, mr)
        m.scan_host(m.packed())
        m.commit()
    assert m.prepare(xyz, origin, res, fr, mr)
    pk = m.packed()
    (a1, b1, s1), t1 = run(m, pk, 1, reps, opts=(("bgk_tables", 0),))
    (a2, b2, s2), t2 = run(m, pk, 1, reps, opts=(("bgk_tables", 1),))
    (a0, b0, s0), t0 = run(m, pk, 0, 1)
    ulp = lambda x, y: np.abs(x.view(np.int32).astype(np.int64) - y.view(np.int32)).max()
    p1, p2, p0 = a1 / (a1 + b1), a2 / (a2 + b2), a0 / (a0 + b0)
    print(f"{name}: leaves {a1.size} tiles {pk.n_test_blk}  t vs r: alpha differ {int((a1 != a2).sum())} (max {ulp(a1, a2)} ulp)  beta differ {int((b1 != b2).sum())} "
          f"(max {ulp(b1, b2)} ulp)  state differ {int((s1 != s2).sum())}  max|dp| {np.abs(p1 - p2).max():.2e};  t vs ordered: max|dp| {np.abs(p2 - p0).max():.2e} "
          f"state differ {int((s0 != s2).sum())};  kernel ms r {np.median(t1):.4f} (min {t1.min():.4f})  t {np.median(t2):.4f} (min {t2.min():.4f})")
    return m, pk


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    xyz, origin = la3dm_amd.load_pcd(os.path.join(root, "tests/golden/data/sim_structured/sim_structured_1.pcd"))
    compare("sim_structured_1 d3", xyz, origin, 0.1, 3, fr=0.5, mr=8.0)
    compare("sim_structured_1 d3, third scan (pruned map)", xyz, origin, 0.1, 3, fr=0.5, mr=8.0, scans=3)
    compare("sim_structured_1 d4", xyz, origin, 0.1, 4, fr=0.5, mr=8.0)
    compare("sim_structured_1 d4, third scan (pruned map)", xyz, origin, 0.1, 4, fr=0.5, mr=8.0, scans=3)
    xyz, origin = la3dm_amd.synthetic_scan(200000, seed=1234)
    m, pk = compare("synthetic 200k d3", xyz, origin, 0.1, 3, reps=20)
    for ab in (1, 2):
        _, t = run(m, pk, 1, 5, opts=(("ablate", ab), ("bgk_tables", 1)))
        print(f"  t ablate {ab}: {np.median(t):.4f} ms")
    m.set_option("ablate", 0)
    compare("synthetic 200k d4", xyz, origin, 0.1, 4, reps=5)
    if "--big" in sys.argv:
        xyz, origin = la3dm_amd.synthetic_scan(1000000, seed=1234)
        compare("synthetic 1M 0.05 d3", xyz, origin, 0.05, 3, reps=3)
