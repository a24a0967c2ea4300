"""HIP path against the restatement in its 'likely reference build' modes (oracle.set_modes(1, 1): Eigen 3.3.7 SSE packet
sin / cos, pcl::VoxelGrid's unstable sort), per map family: max |dp|, share of leaves within 1e-5, structure / state
differences.  The numbers behind DESIGN.md section 4's table and the bounds of the guard tests.
gpurun -- python tools/check/likely_ref.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import la3dm_amd
from oracle import oracle as O

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "data")


def pcd(name, i):
    return la3dm_amd.load_pcd(os.path.join(DATA, name, f"{name}_{i}.pcd"))


def lv_prob(A, B, min_W):
    A, B = A.astype(np.float64), B.astype(np.float64)
    W = np.maximum(A + B, min_W)
    return np.where(A > B, A / (W - B) + (W - A - B) * 0.5 / (W - B), 0.5 * (W - B - A) / (W - A))


def report(tag, a, b, pa, pb):
    same = a["block_key"].size == b["block_key"].size and (a["block_key"] == b["block_key"]).all() and (a["node_key"] == b["node_key"]).all()
    if not same:
        print(f"{tag}: leaf structure differs ({a['block_key'].size} vs {b['block_key'].size})")
        return
    d = np.abs(pa - pb)
    rel = lambda k: float((np.abs(a[k].astype(np.float64) - b[k]) / np.maximum(np.abs(b[k].astype(np.float64)), 1e-3)).max())
    print(f"{tag}: leaves {d.size}  max|dp| {d.max():.3e}  within 1e-5 {float((d <= 1e-5).mean()):.5f}  state differs {int((a['state'] != b['state']).sum())}"
          f"  classified differs {int((a['classified'] != b['classified']).sum())}  max rel dA {rel('A'):.2e} dB {rel('B'):.2e}"
          f"  bit-equal A {float((a['A'] == b['A']).mean()):.4f}")


def bgkl(scans, tag, fr=0.3, mr=8.0, omp=False):
    params = dict(la3dm_amd.L_YAML)
    m = la3dm_amd.BGKLOctoMap(**params, device=0)
    O.set_modes(1, 1, omp=omp)
    try:
        o = O.OracleLMap(**params, omp=omp)
        for k, (xyz, origin) in enumerate(scans):
            m.insert_pointcloud(xyz, origin, 0.1, fr, mr)
            o.insert_pointcloud(xyz, origin, 0.1, fr, mr)
            a, b = m.leaves(), o.leaves()
            report(f"BGK-L {tag} scan {k + 1}", a, b, a["A"].astype(np.float64) / (a["A"].astype(np.float64) + a["B"]),
                   b["A"].astype(np.float64) / (b["A"].astype(np.float64) + b["B"]))
    finally:
        O.set_modes(0, 0, omp=omp)


def lv(scans, tag, res, depth, omp=True):
    params = dict(la3dm_amd.LV_YAML, resolution=res, block_depth=depth)
    m = la3dm_amd.BGKLVOctoMap(**params, device=0)
    O.set_modes(1, 1, omp=omp)
    try:
        o = O.OracleLVMap(**params, omp=omp)
        for k, (xyz, origin) in enumerate(scans):
            m.insert_pointcloud(xyz, origin, res, 0.1, 8.0)
            o.insert_pointcloud(xyz, origin, res, 0.1, 8.0)
            a, b = m.leaves(), o.leaves()
            report(f"BGK-LV {tag} scan {k + 1}", a, b, lv_prob(a["A"], a["B"], params["min_W"]), lv_prob(b["A"], b["B"], params["min_W"]))
    finally:
        O.set_modes(0, 0, omp=omp)


def gp(scans, tag, depth, fr=0.1, mr=8.0, omp=True, gp_mode=1, grid_sort=1):
    """GPOctoMap against the restatement with set_gp_mode(gp_mode): 1 = Eigen 3.3.7's order of operations (+ the voxel
    grid's unstable sort), 2 = double precision throughout (the yardstick)"""
    params = dict(la3dm_amd.GP_YAML, block_depth=depth)
    m = la3dm_amd.GPOctoMap(**params, device=0)
    O.set_gp_mode(gp_mode, omp=omp)
    O.set_modes(0, grid_sort, omp=omp)
    try:
        o = O.OracleGPMap(**params, omp=omp)
        max_ivar = 1.0 / params["min_var"]
        for k, (xyz, origin) in enumerate(scans):
            m.insert_pointcloud(xyz, origin, 0.1, fr, mr)
            o.insert_pointcloud(xyz, origin, 0.1, fr, mr)
            a, b = m.leaves(), o.leaves()
            # leaves()["A"] = m_ivar, ["B"] = ivar (gpoctree_node.cpp:31-34: p = 1 / (1 + exp(-l * m_ivar / max_ivar)))
            pa = 1.0 / (1.0 + np.exp(-params["l"] * a["A"].astype(np.float64) / max_ivar))
            pb = 1.0 / (1.0 + np.exp(-params["l"] * b["A"].astype(np.float64) / max_ivar))
            report(f"GP d{depth} {tag} scan {k + 1} [HIP vs restatement mode {gp_mode}]", a, b, pa, pb)
    finally:
        O.set_gp_mode(0, omp=omp)
        O.set_modes(0, 0, omp=omp)


def gp_modes(scan, tag, depth, fr=0.1, mr=8.0):
    """the restatement's own modes against each other on one scan: FMA chains (0) / Eigen order (1) against double (2)"""
    params = dict(la3dm_amd.GP_YAML, block_depth=depth)
    max_ivar = 1.0 / params["min_var"]
    res = {}
    for mode in (0, 1, 2):
        O.set_gp_mode(mode, omp=True)
        try:
            o = O.OracleGPMap(**params, omp=True)
            o.insert_pointcloud(scan[0], scan[1], 0.1, fr, mr)
            res[mode] = o.leaves()
        finally:
            O.set_gp_mode(0, omp=True)
    for x, y in ((0, 2), (1, 2), (0, 1)):
        a, b = res[x], res[y]
        pa = 1.0 / (1.0 + np.exp(-params["l"] * a["A"].astype(np.float64) / max_ivar))
        pb = 1.0 / (1.0 + np.exp(-params["l"] * b["A"].astype(np.float64) / max_ivar))
        report(f"GP d{depth} {tag} [restatement mode {x} vs mode {y}]", a, b, pa, pb)


if __name__ == "__main__":
    if "gp" in sys.argv:
        gp([pcd("sim_structured", i) for i in range(1, 4)], "sim_structured", 3)
        gp([pcd("sim_structured", i) for i in range(1, 3)], "sim_structured", 4)
        gp([pcd("sim_structured", i) for i in range(1, 3)], "sim_structured", 3, gp_mode=2, grid_sort=0)
        gp([pcd("sim_structured", 1)], "sim_structured", 4, gp_mode=2, grid_sort=0)
        gp_modes(pcd("sim_structured", 1), "sim_structured scan 1", 3)
        gp_modes(pcd("sim_structured", 1), "sim_structured scan 1", 4)
        gp([la3dm_amd.synthetic_scan(50000)], "synthetic 50 k rays (configs[2])", 3, mr=-1.0, grid_sort=0)
        sys.exit(0)
    bgkl([pcd("sim_structured", i) for i in range(1, 4)], "sim_structured")
    bgkl([la3dm_amd.synthetic_scan(50000)], "synthetic 50 k rays", mr=-1.0, omp=True)
    lv([pcd("sim_unstructured", i) for i in range(1, 5)], "sim_unstructured 0.1 m d4", 0.1, 4)
    lv([pcd("sim_unstructured", i) for i in range(1, 4)], "sim_unstructured 0.05 m d5", 0.05, 5)
