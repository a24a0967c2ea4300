"""bgk_sum = 1 (order-free double accumulators) against bgk_sum = 0 (the reference's order): same packed scan, same
initial alpha / beta; prints the differences and the kernel times.  gpurun -- python tools/check/r_check.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import la3dm_amd
from la3dm_amd import _lib


def run(m, pk, mode, reps=1, opts=()):
    m.set_option("bgk_sum", mode)
    for k, v in opts:
        m.set_option(k, v)
    a0, b0 = pk.alpha.copy(), pk.beta.copy()
    m.set_option("time_kernel", 1)
    for _ in range(reps):
        pk.alpha[:] = a0
        pk.beta[:] = b0
        m.scan_host(pk)
    kt = np.zeros(reps + 8, np.float32)
    nk = C.c_uint32()
    _lib.hip().la3dm_kernel_times(m.ctx(), kt.ctypes.data, kt.size, C.byref(nk))
    m.set_option("time_kernel", 0)
    out = pk.alpha.copy(), pk.beta.copy(), pk.state.copy()
    pk.alpha[:] = a0
    pk.beta[:] = b0
    return out, kt[:nk.value]


def compare(name, xyz, origin, res, depth, reps=5, fr=0.5, mr=-1.0):
    params = dict(la3dm_amd.BGK_YAML, resolution=res, block_depth=depth)
    m = la3dm_amd.BGKOctoMap(**params, device=0).set_device_resident(False)
    assert m.prepare(xyz, origin, res, fr, mr)
    pk = m.packed()
    (a0, b0, s0), t0 = run(m, pk, 0, reps)
    (a1, b1, s1), t1 = run(m, pk, 1, reps)
    p0, p1 = a0 / (a0 + b0), a1 / (a1 + b1)
    upd0, upd1 = (s0 & 0x80) != 0, (s1 & 0x80) != 0
    print(f"{name}: leaves {a0.size}  updated {int(upd0.sum())}/{int(upd1.sum())}  classified-flag mismatches {int((upd0 != upd1).sum())}"
          f"  state mismatches {int((s0 != s1).sum())}  max|dp| {np.abs(p0 - p1).max():.3e}  max rel dA {np.abs(a0 - a1).max() / 1:.3e}"
          f"  bit-equal alpha {float((a0 == a1).mean()):.4f}  kernel ms exact {np.median(t0):.4f}  f64-sum {np.median(t1):.4f}")
    return m, pk


def refused(name, m, pk, n=15):
    """the same packed scan fused n times on top of its own result, in both modes: how far the modes drift apart"""
    a0, b0 = pk.alpha.copy(), pk.beta.copy()
    res = []
    for mode in (0, 1):
        m.set_option("bgk_sum", mode)
        pk.alpha[:] = a0
        pk.beta[:] = b0
        for _ in range(n):
            m.scan_host(pk)
        res.append((pk.alpha.copy(), pk.beta.copy()))
    pk.alpha[:] = a0
    pk.beta[:] = b0
    (x0, y0), (x1, y1) = res
    print(f"{name}: {n} fused re-insertions: max|dp| between the modes {np.abs(x0 / (x0 + y0) - x1 / (x1 + y1)).max():.3e}")


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    xyz, origin = la3dm_amd.load_pcd(os.path.join(root, "tests/golden/data/sim_structured/sim_structured_1.pcd"))
    compare("sim_structured_1 d3", xyz, origin, 0.1, 3, fr=0.5, mr=8.0)
    compare("sim_structured_1 d4", xyz, origin, 0.1, 4, fr=0.5, mr=8.0)
    xyz, origin = la3dm_amd.synthetic_scan(200000, seed=1234)
    m, pk = compare("synthetic 200k d3", xyz, origin, 0.1, 3, reps=10)
    refused("synthetic 200k d3", m, pk)
    for ab in (1, 2, 8, 16):
        _, t = run(m, pk, 1, 5, opts=(("ablate", ab),))
        print(f"  f64-sum ablate {ab}: {np.median(t):.4f} ms")
    m.set_option("ablate", 0)
    compare("synthetic 200k d4", xyz, origin, 0.1, 4, reps=5)
    if "--big" in sys.argv:
        xyz, origin = la3dm_amd.synthetic_scan(1000000, seed=1234)
        compare("synthetic 1M 0.05 d3", xyz, origin, 0.05, 3, reps=3)
