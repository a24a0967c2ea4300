"""kernel time of the BGK predict launch on configs[1] (and the big scan with --big): t (tables) vs r, 20 reps; prints
whether t and r agree bit for bit.   gpurun -- python tools/check/t_quick.py [--big]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import la3dm_amd
from r_check import run


def one(name, rays, res, reps):
    xyz, origin = la3dm_amd.synthetic_scan(rays, seed=1234)
    m = la3dm_amd.BGKOctoMap(**dict(la3dm_amd.BGK_YAML, resolution=res), device=0).set_device_resident(False)
    assert m.prepare(xyz, origin, res, 0.5, -1.0)
    pk = m.packed()
    (a1, b1, s1), t1 = run(m, pk, 1, reps, opts=(("bgk_tables", 0),))
    (a2, b2, s2), t2 = run(m, pk, 1, reps, opts=(("bgk_tables", 1),))
    full = pk.c.flags
    pk.c.flags = full & ~4               # without LA3DM_SCAN_FULL_BLOCKS: the kernel that carries the general path
    (a3, b3, s3), t3 = run(m, pk, 1, reps, opts=(("bgk_tables", 1),))
    pk.c.flags = full
    ab = []
    for k in (1, 2):
        _, t = run(m, pk, 1, 5, opts=(("ablate", k), ("bgk_tables", 1)))
        ab.append(float(np.median(t)))
    m.set_option("ablate", 0)
    print(f"{name}: t with the general path compiled in {np.median(t3):.4f} ms (differ {int((a3 != a2).sum())})")
    print(f"{name}: r {np.median(t1):.4f} ms  t {np.median(t2):.4f} ms (min {t2.min():.4f}; no C {ab[0]:.4f}, no B/C {ab[1]:.4f})  "
          f"differ alpha {int((a1 != a2).sum())} beta {int((b1 != b2).sum())} state {int((s1 != s2).sum())}")


if __name__ == "__main__":
    one("200k d3", 200000, 0.1, 20)
    if "--big" in sys.argv:
        one("1M 0.05 d3", 1000000, 0.05, 5)
