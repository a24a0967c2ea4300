#!/usr/bin/env python
"""Generates the committed golden fixtures.  Run in the build container (needs /root/reference
for the first file; the GPU box only reads the committed outputs).

  ref_kat.npz      known answers captured from the REFERENCE's own std-only sources compiled in
                   place (oracle/_ref, see oracle/ref_harness.cpp): block hashing, voxel LUT,
                   leaf order, Occupancy::update sequences, OcTree::prune, R-tree box queries.
  ref_kat_lv.npz   the same for the reference's BGK-LV family (bgklvoctree_node.cpp, bgklvoctree.cpp, bgklvblock.cpp,
                   point6f.cpp; oracle/_ref/libla3dm_ref_lv.so): node update sequences incl. the min_W floor and the
                   UNCERTAIN state, the (A, B) constructor, get_prob / get_var, the 28-bit node key, LUT up to depth 5,
                   leaf order / update / prune rounds of whole blocks, block hashing, point6f's constructors.
  oracle_kat.npz   known answers of the CPU restatement for the parts the reference cannot pin
                   (Eigen/PCL absent): kernel table k(r), per-block predict cases, voxel-grid
                   filter case.  Regression pins + inputs for the GPU parity tests.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

YAML = (0.1, 3, 1.0, 0.2, 0.3, 0.7, 100.0, 0.001, 0.001)


def ref_grid_kat():
    """Block::get_index / get_node / get_point (the RayCaster's primitives) from the reference's compiled sources."""
    R = O.ref()
    assert R is not None, "oracle/_ref not built (needs /root/reference)"
    rng = np.random.default_rng(20260928)
    out = {}
    for depth in (3, 4):
        R.ref_configure(YAML[0], depth, *YAML[2:])
        bs = R.ref_block_size()
        for case, c in enumerate([[0.8, 0.8, 0.0], [-3.6, 12.8, 4.4]]):
            c = (np.array(c) / 0.4 * bs).astype(np.float32)
            b = R.ref_block_new(float(c[0]), float(c[1]), float(c[2]))
            pts = np.concatenate([c + rng.uniform(-0.75 * bs, 0.75 * bs, (300, 3)),       # inside and outside (clamped)
                                  c + (rng.integers(-5, 6, (100, 3)) * YAML[0]),            # on voxel faces
                                  [c, c - bs / 2, c + bs / 2]]).astype(np.float32)
            idx = np.zeros((len(pts), 3), np.int32)
            key = np.zeros(len(pts), np.int32)
            loc = np.zeros((len(pts), 3), np.float32)
            t3, k1, i3 = np.zeros(3, np.float32), C.c_int32(), np.zeros(3, np.int32)
            for i, p in enumerate(pts):
                R.ref_block_grid(b, np.ascontiguousarray(p), i3, C.byref(k1), t3)
                idx[i], key[i], loc[i] = i3, k1.value, t3
            R.ref_block_free(b)
            tag = f"d{depth}_c{case}"
            out[f"{tag}_center"], out[f"{tag}_pts"] = c, pts
            out[f"{tag}_idx"], out[f"{tag}_key"], out[f"{tag}_point"] = idx, key, loc
    np.savez_compressed(os.path.join(HERE, "ref_kat_grid.npz"), **out)
    print("ref_kat_grid.npz:", len(out), "arrays")


def ref_kat():
    R = O.ref()
    assert R is not None, "oracle/_ref not built (needs /root/reference)"
    rng = np.random.default_rng(20260927)
    out = {}
    for depth in (3, 4, 5):
        R.ref_configure(YAML[0], depth, *YAML[2:])
        bs = R.ref_block_size()
        out[f"d{depth}_block_size"] = np.float32(bs)
        # hashing: random points, exact block-face points, negative coordinates
        pts = np.concatenate([rng.uniform(-60, 60, (400, 3)),
                              (rng.integers(-50, 50, (200, 3)) + 0.5) * bs,
                              rng.integers(-50, 50, (100, 3)) * bs,
                              [[0, 0, 0], [0.2, 0.2, 0.2], [0.39, -0.41, 7.45], [-3.75, 12.95, 4.25],
                               [1, 1, 0.0997627]]]).astype(np.float32)
        keys = np.array([R.ref_block_to_hash_key(*p) for p in pts], np.int64)
        centres = np.zeros((len(pts), 3), np.float32)
        ebs = np.zeros((len(pts), 7), np.int64)
        t = np.zeros(3, np.float32)
        e = np.zeros(7, np.int64)
        for i, k in enumerate(keys):
            R.ref_hash_key_to_block(int(k), t)
            centres[i] = t
            R.ref_get_extended_block(int(k), e)
            ebs[i] = e
        out[f"d{depth}_hash_pts"], out[f"d{depth}_hash_keys"] = pts, keys
        out[f"d{depth}_hash_centres"], out[f"d{depth}_eblocks"] = centres, ebs
        # LUT, depth-major
        lut = []
        for d in range(depth):
            for i in range(8 ** d):
                assert R.ref_lut(d, i, t)
                lut.append(t.copy())
        out[f"d{depth}_lut"] = np.asarray(lut, np.float32)
        if depth == 5:
            continue
        # leaf order / positions of a fresh block, then random updates + prune rounds
        cap = 8 ** (depth - 1)
        for case in range(3):
            c = (np.array([[0.8, 0.8, 0.0], [0.2, 0.2, 0.2], [-3.6, 12.8, 4.4]][case]) / 0.4 * bs).astype(np.float32)
            b = R.ref_block_new(float(c[0]), float(c[1]), float(c[2]))
            keys_l = np.zeros(cap, np.int32)
            loc = np.zeros((cap, 3), np.float32)
            sz = np.zeros(cap, np.float32)
            n = R.ref_block_leaves(b, keys_l, loc, sz, cap)
            tag = f"d{depth}_blk{case}"
            out[f"{tag}_center"] = c
            out[f"{tag}_fresh_keys"], out[f"{tag}_fresh_loc"], out[f"{tag}_fresh_size"] = keys_l[:n].copy(), loc[:n].copy(), sz[:n].copy()
            ops = []   # (round, key, ybar, kbar)
            dumps = []
            for rnd in range(4):
                n = R.ref_block_leaves(b, keys_l, loc, sz, cap)
                cur = keys_l[:n].copy()
                # drive whole sibling groups to one state so that pruning triggers
                mode = rng.integers(0, 3, size=n)
                for j, k in enumerate(cur):
                    grp = (int(k) & 0xFFFF) // 8 + (int(k) >> 16) * 100000
                    g = np.random.default_rng(grp + 17 * rnd + 1000 * case).integers(0, 4)
                    if g == 0:
                        yb, kb = 0.0, float(rng.uniform(0.5, 3))       # free
                    elif g == 1:
                        kb = float(rng.uniform(0.5, 3)); yb = kb       # occupied
                    elif g == 2:
                        kb = float(rng.uniform(0.0, 2)); yb = kb * float(rng.uniform(0, 1))
                    else:
                        continue
                    R.ref_block_update(b, int(k), yb, kb)
                    ops.append((rnd, int(k), yb, kb))
                pr = R.ref_block_prune(b)
                n = R.ref_block_leaves(b, keys_l, loc, sz, cap)
                A = C.c_float(); B = C.c_float(); S = C.c_uint8(); P = C.c_float(); V = C.c_float()
                rows = []
                for k in keys_l[:n]:
                    assert R.ref_block_node(b, int(k), C.byref(A), C.byref(B), C.byref(S), C.byref(P), C.byref(V))
                    rows.append((int(k), A.value, B.value, S.value, P.value, V.value))
                dumps.append((rnd, pr, np.asarray(rows, np.float64), loc[:n].copy(), sz[:n].copy()))
            out[f"{tag}_ops"] = np.asarray(ops, np.float64)
            for rnd, pr, rows, lc, s_ in dumps:
                out[f"{tag}_r{rnd}_pruned"] = np.int32(pr)
                out[f"{tag}_r{rnd}_leaves"] = rows
                out[f"{tag}_r{rnd}_loc"] = lc
                out[f"{tag}_r{rnd}_size"] = s_
            R.ref_block_free(b)
    # node update sequences
    R.ref_configure(*YAML)
    yb = np.concatenate([[0, 0.5, 0, 3, 2.5], rng.uniform(0, 1, 200)]).astype(np.float32)
    kb = np.concatenate([[1e-9, 0.75, 2, 3, 2.5], rng.uniform(0, 1, 200)]).astype(np.float32)
    kb[5:] = np.maximum(kb[5:], yb[5:])
    n = len(yb)
    A = np.zeros(n, np.float32); B = A.copy(); S = np.zeros(n, np.uint8); P = A.copy(); V = A.copy()
    R.ref_node_sequence(yb, kb, n, A, B, S, P, V)
    out.update(node_ybar=yb, node_kbar=kb, node_A=A, node_B=B, node_state=S, node_prob=P, node_var=V)
    # R-tree closed-box rule: points on faces/corners/1-ulp off, queried through the reference's
    # own RTree<int,float,3,float> with the box arithmetic of get_gp_points_in_bbox
    bs = R.ref_block_size()
    grid = (rng.integers(-6, 6, (300, 3)) + rng.choice([0.0, 0.5, -0.5], (300, 3))) * bs
    pts = np.concatenate([rng.uniform(-2.5, 2.5, (2000, 3)), grid,
                          np.nextafter((grid[:100]).astype(np.float32), np.float32(np.inf)),
                          np.nextafter((grid[100:200]).astype(np.float32), np.float32(-np.inf)),
                          [[0, 0, 0], [0.2, 0.2, 0.2], [0.2, 0, 0], [0.5, 0.5, 0.5], [-0.2, 0.1, 0.1]]]).astype(np.float32)
    tree = R.ref_rtree_new(np.ascontiguousarray(pts), len(pts))
    qkeys = sorted(set(int(R.ref_block_to_hash_key(*p)) for p in pts))
    ids = np.zeros(len(pts), np.int32)
    qk, qoff, qids = [], [0], []
    for k in qkeys:
        n = R.ref_rtree_block_query(tree, k, ids, len(ids))
        qk.append(k)
        qids.extend(sorted(ids[:n].tolist()))
        qoff.append(len(qids))
    R.ref_rtree_free(tree)
    out.update(rtree_pts=pts, rtree_keys=np.asarray(qk, np.int64), rtree_off=np.asarray(qoff, np.int64),
               rtree_ids=np.asarray(qids, np.int32))
    np.savez_compressed(os.path.join(HERE, "ref_kat.npz"), **out)
    print("ref_kat.npz:", len(out), "arrays")


LV_YAML = (0.1, 5, 0.1, 0.2, 0.3, 0.7, 0.2, 0.001, 0.001)     # config/methods/bgklvoctomap.yaml (+ original_size, min_W)


def ref_kat_lv():
    R = O.ref_lv()
    assert R is not None, "oracle/_ref/libla3dm_ref_lv.so not built (needs /root/reference)"
    rng = np.random.default_rng(20260929)
    out = {}
    R.ref_configure_lv(1, 0.001)
    # 28-bit node keys
    dk = np.array([(d, i) for d in range(6) for i in (0, 1, 7, 8 ** d - 1 if d else 0, (8 ** d) // 3)], np.int64)
    keys = np.array([R.ref_node_to_hash_key(int(d), int(i)) for d, i in dk], np.int32)
    back = np.zeros((len(keys), 2), np.int64)
    d_, i_ = C.c_int32(), C.c_uint32()
    for j, k in enumerate(keys):
        R.ref_hash_key_to_node(int(k), C.byref(d_), C.byref(i_))
        back[j] = (d_.value, i_.value)
    out.update(key_depth_index=dk, key_value=keys, key_back=back)
    t = np.zeros(3, np.float32)
    e = np.zeros(7, np.int64)
    for res, depth in ((0.1, 3), (0.1, 4), (0.05, 5)):
        R.ref_configure(res, depth, *LV_YAML[2:])
        bs = R.ref_block_size()
        tag = f"d{depth}"
        out[f"{tag}_resolution"], out[f"{tag}_block_size"] = np.float32(res), np.float32(bs)
        pts = np.concatenate([rng.uniform(-40, 40, (300, 3)), (rng.integers(-50, 50, (150, 3)) + 0.5) * bs,
                              rng.integers(-50, 50, (80, 3)) * bs, [[0, 0, 0], [0.39, -0.41, 7.45], [-3.75, 12.95, 4.25]]]).astype(np.float32)
        hk = np.array([R.ref_block_to_hash_key(*p) for p in pts], np.int64)
        centres = np.zeros((len(pts), 3), np.float32)
        ebs = np.zeros((len(pts), 7), np.int64)
        for i, k in enumerate(hk):
            R.ref_hash_key_to_block(int(k), t)
            centres[i] = t
            R.ref_get_extended_block(int(k), e)
            ebs[i] = e
        out[f"{tag}_hash_pts"], out[f"{tag}_hash_keys"], out[f"{tag}_hash_centres"], out[f"{tag}_eblocks"] = pts, hk, centres, ebs
        lut = []
        for d in range(depth):
            for i in range(8 ** d):
                assert R.ref_lut(d, i, t)
                lut.append(t.copy())
        out[f"{tag}_lut"] = np.asarray(lut, np.float32)
        # whole blocks: leaf order, update rounds that drive sibling groups to one state (incl. UNCERTAIN), prune
        cap = 8 ** (depth - 1)
        for case in range(2):
            c = (np.array([[0.8, 0.8, 0.0], [-3.6, 12.8, 4.4]][case]) / 0.4 * bs).astype(np.float32)
            b = R.ref_block_new(float(c[0]), float(c[1]), float(c[2]))
            keys_l = np.zeros(cap, np.int32)
            loc = np.zeros((cap, 3), np.float32)
            sz = np.zeros(cap, np.float32)
            n = R.ref_block_leaves(b, keys_l, loc, sz, cap)
            bt = f"{tag}_blk{case}"
            out[f"{bt}_center"] = c
            out[f"{bt}_fresh_keys"], out[f"{bt}_fresh_loc"], out[f"{bt}_fresh_size"] = keys_l[:n].copy(), loc[:n].copy(), sz[:n].copy()
            ops = []
            for rnd in range(4):
                n = R.ref_block_leaves(b, keys_l, loc, sz, cap)
                for k in keys_l[:n].copy():
                    kd, ki = (int(k) >> 28) & 0xF, int(k) & 0xFFFFFFF
                    g = np.random.default_rng((ki // 8) + kd * 10 ** 7 + 17 * rnd + 1000 * case).integers(0, 5)
                    if g == 0:
                        yb, kb = 0.0, float(rng.uniform(0.5, 3))                  # free
                    elif g == 1:
                        kb = float(rng.uniform(0.5, 3)); yb = kb                  # occupied
                    elif g == 2:
                        kb = float(rng.uniform(0.0, 2)); yb = kb * float(rng.uniform(0, 1))
                    elif g == 3:
                        kb = float(rng.uniform(0.002, 0.05)); yb = kb * 0.5       # p = 0.5 with little mass: UNCERTAIN (var 0.25 > 0.2)
                    else:
                        continue
                    R.ref_block_update(b, int(k), yb, kb)
                    ops.append((rnd, int(k), yb, kb))
                pr = R.ref_block_prune(b)
                n = R.ref_block_leaves(b, keys_l, loc, sz, cap)
                A = C.c_float(); B = C.c_float(); S = C.c_uint8(); P = C.c_float(); V = C.c_float()
                rows = []
                for k in keys_l[:n]:
                    assert R.ref_block_node(b, int(k), C.byref(A), C.byref(B), C.byref(S), C.byref(P), C.byref(V))
                    rows.append((int(k), A.value, B.value, S.value, P.value, V.value))
                out[f"{bt}_r{rnd}_pruned"] = np.int32(pr)
                out[f"{bt}_r{rnd}_leaves"] = np.asarray(rows, np.float64)
                out[f"{bt}_r{rnd}_loc"], out[f"{bt}_r{rnd}_size"] = loc[:n].copy(), sz[:n].copy()
            out[f"{bt}_ops"] = np.asarray(ops, np.float64)
            R.ref_block_free(b)
    # node sequences: the YAML's statics and the constructor defaults of bgklvoctomap.cpp:21-32 (min_W 0.1: the floor is active)
    for name, cfg, min_w in (("yaml", LV_YAML, 0.001), ("ctor", (0.1, 4, 1.0, 1.0, 0.3, 0.7, 1.0, 1.0, 1.0), 0.1),
                             ("floor", (0.1, 4, 1.0, 1.0, 0.3, 0.7, 0.2, 0.001, 0.001), 0.1)):
        R.ref_configure(*cfg)
        R.ref_configure_lv(1, min_w)
        for seq in range(4):
            n = 160
            scale = [1.0, 0.01, 0.2, 3.0][seq]
            kb = (rng.uniform(0, 1, n) * scale).astype(np.float32)
            # seq 0 mostly free evidence, seq 2 mostly occupied, seq 1 tiny masses around min_W, seq 3 mixed
            frac = [rng.choice([0.0, 0.0, 0.0, 0.05], n), rng.choice([0.0, 1.0, 0.5, 0.3], n), rng.choice([1.0, 1.0, 1.0, 0.9], n),
                    rng.choice([0.0, 1.0, 0.5, 0.3], n)][seq]
            yb = (kb * frac).astype(np.float32)
            if seq == 1:
                kb[:8], yb[:8] = [1e-9, 0.0015, 0.004, 0.0, 0.02, 0.001, 0.05, 0.09], [0, 0.00075, 0.002, 0, 0.01, 0.001, 0.025, 0]
            A = np.zeros(n, np.float32); B = A.copy(); S = np.zeros(n, np.uint8); P = A.copy(); V = A.copy()
            R.ref_node_sequence(yb, kb, n, A, B, S, P, V)
            t_ = f"node_{name}{seq}"
            out.update({f"{t_}_ybar": yb, f"{t_}_kbar": kb, f"{t_}_A": A, f"{t_}_B": B, f"{t_}_state": S, f"{t_}_prob": P, f"{t_}_var": V})
        ab = np.concatenate([rng.uniform(0, 2, (200, 2)), rng.uniform(0, 0.05, (200, 2)), [[0, 0], [5, 0], [0, 5], [0.05, 0.04], [0.0004, 0.0005]]]).astype(np.float32)
        res_ = np.zeros((len(ab), 5), np.float64)
        mA, mB, S1, P1, V1 = C.c_float(), C.c_float(), C.c_uint8(), C.c_float(), C.c_float()
        for i, (a_, b_) in enumerate(ab):
            R.ref_node_ctor(float(a_), float(b_), C.byref(mA), C.byref(mB), C.byref(S1), C.byref(P1), C.byref(V1))
            res_[i] = (mA.value, mB.value, S1.value, P1.value, V1.value)
        out[f"ctor_{name}_ab"], out[f"ctor_{name}_out"] = ab, res_
        out[f"cfg_{name}"] = np.asarray(list(cfg) + [min_w], np.float64)
    # point6f
    a3, b3 = rng.uniform(-5, 5, 3).astype(np.float32), rng.uniform(-5, 5, 3).astype(np.float32)
    o = [np.zeros(6, np.float32) for _ in range(4)]
    R.ref_point6f(a3, b3, *o)
    out.update(p6_a=a3, p6_b=b3, p6_from_point=o[0], p6_from_pair=o[1], p6_from_xyz=o[2], p6_start_end=o[3])
    np.savez_compressed(os.path.join(HERE, "ref_kat_lv.npz"), **out)
    print("ref_kat_lv.npz:", len(out), "arrays")


def oracle_kat():
    rng = np.random.default_rng(7)
    out = {}
    r = np.concatenate([np.linspace(0, 1.25, 5001), rng.uniform(0.9, 1.0, 3000), [0.0, 0.5, 0.9, 0.99, 1.0, 1.05]]).astype(np.float32)
    out["kernel_r"] = r
    out["kernel_k_sf1"] = O.kernel(r, 1.0)
    out["kernel_k_sf01"] = O.kernel(r, 0.1)
    # per-block predict cases: rim pairs, empty-ish neighbours, duplicates
    cases = []
    for ci in range(6):
        M = [64, 64, 8, 1, 64, 37][ci]
        N = [16, 56, 3, 1, 320, 5][ci]
        xs = rng.uniform(-0.2, 0.2, (M, 3)).astype(np.float32)
        x = rng.uniform(-0.4, 0.4, (N, 3)).astype(np.float32)
        if ci == 2:
            x[1] = x[0]                       # duplicate training point
            x[2] = xs[0] + np.float32(0.2) * np.array([1, 0, 0], np.float32)   # exactly at the rim
        y = (rng.uniform(0, 1, N) < 0.3).astype(np.float32)
        yb, kb = O.bgk_predict(1.0, 0.2, xs, x, y)
        out[f"pred{ci}_xs"], out[f"pred{ci}_x"], out[f"pred{ci}_y"] = xs, x, y
        out[f"pred{ci}_ybar"], out[f"pred{ci}_kbar"] = yb, kb
    # voxel grid (restated PCL semantics)
    pts = rng.uniform(-1, 1, (500, 3)).astype(np.float32)
    out["vg_in"] = pts
    out["vg_out_0p1"] = O.voxel_grid(pts, 0.1)
    out["vg_out_0p25"] = O.voxel_grid(pts, 0.25)
    np.savez_compressed(os.path.join(HERE, "oracle_kat.npz"), **out)
    print("oracle_kat.npz:", len(out), "arrays")


def scan_kat():
    """Full-scan leaf dumps of the CPU restatement (regression pins of the whole path; the GPU tests compare the HIP
    path against these files directly, besides the live oracle): sim_structured scan 1 (BASELINE configs[0]) for
    BGKOctoMap at depth 3 and 4, GPOctoMap, BGKLOctoMap; sim_unstructured scan 1 for BGKLVOctoMap at 0.05 m."""
    sys.path.insert(0, os.path.join(ROOT))
    import la3dm_amd
    out = {}
    xyz, origin = la3dm_amd.load_pcd(os.path.join(HERE, "data", "sim_structured", "sim_structured_1.pcd"))

    def put(tag, lv, keys=("block_key", "node_key", "A", "B", "state", "classified")):
        for k in keys:
            out[f"{tag}_{k}"] = lv[k]

    for depth in (3, 4):
        o = O.OracleMap(**dict(O.BGK_YAML, block_depth=depth))
        o.insert_pointcloud(xyz, origin, 0.1, 0.5, 8.0)
        put(f"bgk_d{depth}", o.leaves())
    o = O.OracleGPMap(**O.GP_YAML)
    o.insert_pointcloud(xyz[::3], origin, 0.1, 0.5, 8.0)
    put("gp_d3", o.leaves())
    o = O.OracleLMap(**O.L_YAML)
    o.insert_pointcloud(xyz, origin, 0.1, 0.3, 8.0)
    put("bgkl_d3", o.leaves())
    xyz_u, origin_u = la3dm_amd.load_pcd(os.path.join(HERE, "data", "sim_unstructured", "sim_unstructured_1.pcd"))
    o = O.OracleLVMap(**dict(O.LV_YAML, resolution=0.05))
    o.insert_pointcloud(xyz_u, origin_u, 0.05, 0.1, 8.0)
    put("lv_d5", o.leaves())
    np.savez_compressed(os.path.join(HERE, "scan_kat.npz"), **out)
    print("scan_kat.npz:", len(out), "arrays,", {k: int(v.shape[0]) for k, v in out.items() if k.endswith("_A")})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "grid":   # added later: does not disturb the older fixtures
        ref_grid_kat()
    elif len(sys.argv) > 1 and sys.argv[1] == "scan":
        scan_kat()
    elif len(sys.argv) > 1 and sys.argv[1] == "lv":     # round 4: the reference's compiled BGK-LV node / tree / block
        ref_kat_lv()
    else:
        ref_kat()
        oracle_kat()
        ref_grid_kat()
        scan_kat()
