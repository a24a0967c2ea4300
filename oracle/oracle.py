"""ctypes front end of the CPU parity oracle (oracle/la3dm_oracle.cpp) and of the
compiled reference layer (oracle/_ref/libla3dm_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (la3dm_amd) never imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")

STATS_FIELDS = ["n_hits", "n_frees", "n_bbox_blocks", "n_train_blocks", "n_test_blocks", "voxel_updates",
                "update_calls", "pair_evals", "train_reads", "t_frontend", "t_partition", "t_predict", "t_prune",
                "t_total"]


def build(force=False):
    """Compile the oracle (and, when /root/reference is present, oracle/_ref)."""
    need = force or not (os.path.exists(os.path.join(_HERE, "liboracle.so"))
                         and os.path.exists(os.path.join(_HERE, "liboracle_omp.so")))
    if need or max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("la3dm_oracle.cpp", "la3dm_oracle_lv.cpp")) > \
            os.path.getmtime(os.path.join(_HERE, "liboracle.so")):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so", "liboracle_omp.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and (force or not all(os.path.exists(
            os.path.join(_HERE, "_ref", f)) for f in ("libla3dm_ref.so", "libla3dm_ref_lv.so"))):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def _load(name):
    lib = C.CDLL(os.path.join(_HERE, name))
    lib.orc_map_create.restype = C.c_void_p
    lib.orc_map_create.argtypes = [C.c_float, C.c_int] + [C.c_float] * 7
    lib.orc_gp_map_create.restype = C.c_void_p
    lib.orc_gp_map_create.argtypes = [C.c_float, C.c_int] + [C.c_float] * 9
    lib.orc_gp_train_predict.argtypes = [C.c_void_p, f32p, f32p, C.c_int, f32p, C.c_int, f32p, f32p, f32p, f32p]
    lib.orc_gp_node_update.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                                       C.c_float, C.c_float]
    lib.orc_gp_node_prob.restype = C.c_float
    lib.orc_gp_node_prob.argtypes = [C.c_void_p, C.c_float]
    lib.orc_lv_map_create.restype = C.c_void_p
    lib.orc_lv_map_create.argtypes = [C.c_float, C.c_int] + [C.c_float] * 7 + [C.c_int, C.c_float]
    lib.orc_lv_map_destroy.argtypes = [C.c_void_p]
    lib.orc_lv_insert_pointcloud.argtypes = [C.c_void_p, f32p, C.c_int64, f32p, C.c_float, C.c_float, C.c_float]
    lib.orc_lv_stats.argtypes = [C.c_void_p, f64p]
    lib.orc_lv_training_data.restype = C.c_int64
    lib.orc_lv_training_data.argtypes = [C.c_void_p, f32p, C.c_int64, f32p, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                         C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.orc_lv_node_ctor.argtypes = [C.c_void_p, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     C.POINTER(C.c_uint8)]
    lib.orc_lv_block_to_hash_key.restype = C.c_int64
    lib.orc_lv_block_to_hash_key.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    lib.orc_lv_hash_key_to_block.argtypes = [C.c_void_p, C.c_int64, f32p]
    lib.orc_lv_lut.restype = C.c_int
    lib.orc_lv_lut.argtypes = [C.c_void_p, C.c_int, C.c_uint32, f32p]
    lib.orc_lv_block_new.restype = C.c_void_p
    lib.orc_lv_block_new.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    lib.orc_lv_block_free.argtypes = [C.c_void_p]
    lib.orc_lv_block_leaves.restype = C.c_int
    lib.orc_lv_block_leaves.argtypes = [C.c_void_p, C.c_void_p, i32p, f32p, f32p, C.c_int]
    lib.orc_lv_block_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float]
    lib.orc_lv_block_prune.restype = C.c_int
    lib.orc_lv_block_prune.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_lv_block_node.restype = C.c_int
    lib.orc_lv_block_node.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
    lib.orc_lv_seg_dist.restype = C.c_float
    lib.orc_lv_seg_dist.argtypes = [f32p, f32p, f32p]
    lib.orc_lv_kernel.restype = C.c_float
    lib.orc_lv_kernel.argtypes = [C.c_float, C.c_float, C.c_float]
    lib.orc_lv_node_update.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                                       C.c_float, C.c_float]
    lib.orc_lv_node_prob.restype = C.c_float
    lib.orc_lv_node_prob.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.orc_lv_node_var.restype = C.c_float
    lib.orc_lv_node_var.argtypes = [C.c_void_p, C.c_float, C.c_float]
    lib.orc_lv_block_count.restype = C.c_int64
    lib.orc_lv_block_count.argtypes = [C.c_void_p]
    lib.orc_lv_dump_leaves.restype = C.c_int64
    lib.orc_lv_dump_leaves.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.orc_map_destroy.argtypes = [C.c_void_p]
    lib.orc_block_size.restype = C.c_float
    lib.orc_block_size.argtypes = [C.c_void_p]
    lib.orc_block_to_hash_key.restype = C.c_int64
    lib.orc_block_to_hash_key.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    lib.orc_hash_key_to_block.argtypes = [C.c_void_p, C.c_int64, f32p]
    lib.orc_get_extended_block.argtypes = [C.c_void_p, C.c_int64, i64p]
    lib.orc_lut_count.restype = C.c_int
    lib.orc_lut_count.argtypes = [C.c_void_p, C.c_int]
    lib.orc_lut.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p]
    lib.orc_kernel.restype = C.c_float
    lib.orc_kernel.argtypes = [C.c_float, C.c_float]
    lib.orc_kernel_array.argtypes = [f32p, C.c_int, C.c_float, f32p]
    lib.orc_kernel_raw.restype = C.c_float
    lib.orc_kernel_raw.argtypes = [C.c_float, C.c_float]
    lib.orc_kernel_max_over.restype = C.c_float
    lib.orc_kernel_max_over.argtypes = [C.c_float, C.c_float, C.c_float]
    lib.orc_bgk_predict.argtypes = [C.c_float, C.c_float, f32p, C.c_int, f32p, f32p, C.c_int, f32p, f32p]
    lib.orc_node_update.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                                    C.c_float, C.c_float]
    lib.orc_node_var.restype = C.c_float
    lib.orc_node_var.argtypes = [C.c_float, C.c_float]
    lib.orc_node_prob.restype = C.c_float
    lib.orc_node_prob.argtypes = [C.c_float, C.c_float]
    lib.orc_voxel_grid.restype = C.c_int
    lib.orc_voxel_grid.argtypes = [f32p, C.c_int, C.c_float, f32p]
    lib.orc_beam_sample.restype = C.c_int
    lib.orc_beam_sample.argtypes = [f32p, f32p, C.c_float, f32p, C.c_int]
    lib.orc_get_training_data.restype = C.c_int64
    lib.orc_get_training_data.argtypes = [f32p, C.c_int64, f32p, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                          C.c_int64]
    lib.orc_insert_pointcloud.argtypes = [C.c_void_p, f32p, C.c_int64, f32p, C.c_float, C.c_float, C.c_float]
    lib.orc_l_map_create.restype = C.c_void_p
    lib.orc_l_map_create.argtypes = [C.c_float, C.c_int] + [C.c_float] * 7
    lib.orc_l_insert_pointcloud.argtypes = [C.c_void_p, f32p, C.c_int64, f32p, C.c_float, C.c_float, C.c_float]
    lib.orc_l_training_data.restype = C.c_int64
    lib.orc_l_training_data.argtypes = [f32p, C.c_int64, f32p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.orc_l_seg_dist.restype = C.c_float
    lib.orc_l_seg_dist.argtypes = [f32p, f32p, f32p]
    lib.orc_insert_xy.argtypes = [C.c_void_p, f32p, C.c_int64]
    lib.orc_insert_training_data.argtypes = [C.c_void_p, f32p, C.c_int64]
    lib.orc_stats.argtypes = [C.c_void_p, f64p]
    lib.orc_num_threads.restype = C.c_int
    lib.orc_set_modes.argtypes = [C.c_int, C.c_int]
    lib.orc_set_sum_mode.argtypes = [C.c_int]
    lib.orc_trig_array.argtypes = [f32p, C.c_int, C.c_int, f32p]
    lib.orc_set_gp_mode.argtypes = [C.c_int]
    lib.orc_block_count.restype = C.c_int64
    lib.orc_block_count.argtypes = [C.c_void_p]
    lib.orc_leaf_count.restype = C.c_int64
    lib.orc_leaf_count.argtypes = [C.c_void_p]
    lib.orc_block_grid.argtypes = [C.c_void_p, f32p, f32p, i32p, C.POINTER(C.c_int32), f32p]
    lib.orc_raycast.restype = C.c_int64
    lib.orc_raycast.argtypes = [C.c_void_p, f32p, f32p, f32p, i64p, i32p, u8p, f32p, f32p, u8p, C.c_int64]
    lib.orc_export_cells.restype = C.c_int64
    lib.orc_export_cells.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, f32p, f32p, i32p, C.c_int64]
    lib.orc_get_bbox.argtypes = [C.c_void_p, f32p, f32p]
    lib.orc_height_map_color.argtypes = [C.c_double, f32p]
    lib.orc_dump_leaves.restype = C.c_int64
    lib.orc_dump_leaves.argtypes = [C.c_void_p, i64p, i32p, f32p, f32p, f32p, f32p, u8p, u8p, C.c_int64]
    lib.orc_block_new.restype = C.c_void_p
    lib.orc_block_new.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
    lib.orc_block_free.argtypes = [C.c_void_p]
    lib.orc_block_leaves.restype = C.c_int
    lib.orc_block_leaves.argtypes = [C.c_void_p, C.c_void_p, i32p, f32p, C.c_int]
    lib.orc_block_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float]
    lib.orc_block_prune.restype = C.c_int
    lib.orc_block_prune.argtypes = [C.c_void_p, C.c_void_p]
    lib.orc_block_node.restype = C.c_int
    lib.orc_block_node.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                   C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
    lib.orc_box_query.restype = C.c_int
    lib.orc_box_query.argtypes = [C.c_void_p, f32p, C.c_int, C.c_int64, i32p, C.c_int]
    return lib


_libs = {}


def lib(omp=False):
    name = "liboracle_omp.so" if omp else "liboracle.so"
    if name not in _libs:
        build()
        _libs[name] = _load(name)
    return _libs[name]


BGK_YAML = dict(resolution=0.1, block_depth=3, sf2=1.0, ell=0.2, free_thresh=0.3, occupied_thresh=0.7,
                var_thresh=100.0, prior_A=0.001, prior_B=0.001)


class OracleMap:
    """CPU BGKOctoMap restatement (same constructor argument order as the reference's
    BGKOctoMap, include/bgkoctomap/bgkoctomap.h:50-58)."""

    def __init__(self, resolution=0.1, block_depth=3, sf2=1.0, ell=0.2, free_thresh=0.3, occupied_thresh=0.7,
                 var_thresh=100.0, prior_A=0.001, prior_B=0.001, omp=False):
        self.L = lib(omp)
        self.h = self.L.orc_map_create(resolution, block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh,
                                       prior_A, prior_B)
        self.block_depth = block_depth

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_map_destroy(self.h)
            self.h = None

    @property
    def block_size(self):
        return self.L.orc_block_size(self.h)

    def block_to_hash_key(self, x, y, z):
        return self.L.orc_block_to_hash_key(self.h, x, y, z)

    def hash_key_to_block(self, key):
        o = np.zeros(3, np.float32)
        self.L.orc_hash_key_to_block(self.h, key, o)
        return o

    def get_extended_block(self, key):
        o = np.zeros(7, np.int64)
        self.L.orc_get_extended_block(self.h, key, o)
        return o

    def lut(self):
        """list over depth of (8^d, 3) float32 arrays"""
        out = []
        for d in range(self.block_depth):
            n = self.L.orc_lut_count(self.h, d)
            a = np.zeros((n, 3), np.float32)
            t = np.zeros(3, np.float32)
            for i in range(n):
                self.L.orc_lut(self.h, d, i, t)
                a[i] = t
            out.append(a)
        return out

    def insert_pointcloud(self, xyz, origin, ds_resolution, free_res=2.0, max_range=-1.0):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        origin = np.ascontiguousarray(origin, np.float32)
        self.L.orc_insert_pointcloud(self.h, xyz, xyz.shape[0], origin, ds_resolution, free_res, max_range)

    def insert_training_data(self, xyzy):
        """BGKOctoMap::insert_training_data: the scan stages from a labelled set, updates not gated on kbar"""
        xyzy = np.ascontiguousarray(xyzy, np.float32).reshape(-1, 4)
        self.L.orc_insert_training_data(self.h, xyzy, xyzy.shape[0])

    def insert_xy(self, xyzy):
        xyzy = np.ascontiguousarray(xyzy, np.float32).reshape(-1, 4)
        self.L.orc_insert_xy(self.h, xyzy, xyzy.shape[0])

    def stats(self):
        a = np.zeros(len(STATS_FIELDS), np.float64)
        self.L.orc_stats(self.h, a)
        return dict(zip(STATS_FIELDS, a.tolist()))

    def raycast(self, start, end, cap=4096):
        """RayCaster restatement: one row per next() call"""
        s3, e3 = np.ascontiguousarray(start, np.float32), np.ascontiguousarray(end, np.float32)
        out = dict(p=np.zeros((cap, 3), np.float32), block_key=np.zeros(cap, np.int64), node_key=np.zeros(cap, np.int32),
                   valid=np.zeros(cap, np.uint8), A=np.zeros(cap, np.float32), B=np.zeros(cap, np.float32),
                   state=np.zeros(cap, np.uint8))
        n = self.L.orc_raycast(self.h, s3, e3, out["p"], out["block_key"], out["node_key"], out["valid"], out["A"],
                               out["B"], out["state"], cap)
        n = min(int(n), cap)
        return {k: v[:n] for k, v in out.items()}

    def get_bbox(self):
        lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self.L.orc_get_bbox(self.h, lo, hi)
        return lo, hi

    def export_cells(self, state="occupied", original_size=True, min_z=0.0, max_z=0.0):
        """the static node's publish loop (cube lists): cells (n, 4), rgba (n, 4), level (n,)"""
        st = {"occupied": 1, "free": 0}[state]
        e = np.zeros(4, np.float32)
        n = int(self.L.orc_export_cells(self.h, st, int(bool(original_size)), min_z, max_z, e, e, np.zeros(1, np.int32), 0))
        out = dict(cells=np.zeros((max(n, 1), 4), np.float32), rgba=np.zeros((max(n, 1), 4), np.float32),
                   level=np.zeros(max(n, 1), np.int32))
        m = int(self.L.orc_export_cells(self.h, st, int(bool(original_size)), min_z, max_z, out["cells"], out["rgba"],
                                        out["level"], n))
        assert m == n
        return {k: v[:n] for k, v in out.items()}

    def leaves(self):
        n = self.L.orc_leaf_count(self.h)
        out = dict(block_key=np.zeros(n, np.int64), node_key=np.zeros(n, np.int32), loc=np.zeros((n, 3), np.float32),
                   size=np.zeros(n, np.float32), A=np.zeros(n, np.float32), B=np.zeros(n, np.float32),
                   state=np.zeros(n, np.uint8), classified=np.zeros(n, np.uint8))
        m = self.L.orc_dump_leaves(self.h, out["block_key"], out["node_key"], out["loc"], out["size"], out["A"],
                                   out["B"], out["state"], out["classified"], n)
        assert m == n
        return out


L_YAML = dict(resolution=0.1, block_depth=3, sf2=0.1, ell=0.2, free_thresh=0.3, occupied_thresh=0.7, var_thresh=0.15,
              prior_A=0.001, prior_B=0.001)   # config/methods/bgkloctomap.yaml (free_resolution 0.3, ds_resolution 0.1)


class OracleLMap(OracleMap):
    """CPU BGKLOctoMap restatement (block-level BGK with free-space line segments, include/bgkloctomap/bgkloctomap.h)."""

    def __init__(self, resolution=0.1, block_depth=3, sf2=0.1, ell=0.2, free_thresh=0.3, occupied_thresh=0.7,
                 var_thresh=0.15, prior_A=0.001, prior_B=0.001, omp=False):
        self.L = lib(omp)
        self.h = self.L.orc_l_map_create(resolution, block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh,
                                         prior_A, prior_B)
        self.block_depth = block_depth

    def insert_pointcloud(self, xyz, origin, ds_resolution, free_res=2.0, max_range=-1.0):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        origin = np.ascontiguousarray(origin, np.float32)
        self.L.orc_l_insert_pointcloud(self.h, xyz, xyz.shape[0], origin, ds_resolution, free_res, max_range)


def l_training_data(xyz, origin, ds_resolution, free_res, max_range):
    """BGKLOctoMap training set: samples (x, y, z, ray index or -1), rays (6 floats)"""
    L = lib()
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    origin = np.ascontiguousarray(origin, np.float32)
    nr = C.c_int64()
    n = L.orc_l_training_data(xyz, xyz.shape[0], origin, ds_resolution, free_res, max_range, None, 0, None, 0, C.byref(nr))
    xy, rays = np.zeros((n, 4), np.float32), np.zeros((nr.value, 6), np.float32)
    L.orc_l_training_data(xyz, xyz.shape[0], origin, ds_resolution, free_res, max_range, xy.ctypes.data, n,
                          rays.ctypes.data, nr.value, C.byref(nr))
    return xy, rays


GP_YAML = dict(resolution=0.1, block_depth=3, sf2=1.0, ell=1.0, noise=0.01, l=100.0, min_var=0.001, max_var=1000.0,
               max_known_var=0.02, free_thresh=0.3, occupied_thresh=0.7)   # config/methods/gpoctomap.yaml


class OracleGPMap(OracleMap):
    """CPU GPOctoMap restatement (constructor argument order of include/gpoctomap/gpoctomap.h)."""

    def __init__(self, resolution=0.1, block_depth=4, sf2=1.0, ell=1.0, noise=0.01, l=100.0, min_var=0.001,
                 max_var=1000.0, max_known_var=0.02, free_thresh=0.3, occupied_thresh=0.7, omp=False):
        self.L = lib(omp)
        self.h = self.L.orc_gp_map_create(resolution, block_depth, sf2, ell, noise, l, min_var, max_var,
                                          max_known_var, free_thresh, occupied_thresh)
        self.block_depth = block_depth

    def train_predict(self, x, y, xs):
        x = np.ascontiguousarray(x, np.float32).reshape(-1, 3)
        y = np.ascontiguousarray(y, np.float32)
        xs = np.ascontiguousarray(xs, np.float32).reshape(-1, 3)
        n, m = x.shape[0], xs.shape[0]
        alpha, Lm = np.zeros(n, np.float32), np.zeros((n, n), np.float32)
        mu, var = np.zeros(m, np.float32), np.zeros(m, np.float32)
        self.L.orc_gp_train_predict(self.h, x, y, n, xs, m, alpha, Lm, mu, var)
        return alpha, Lm, mu, var


LV_YAML = dict(resolution=0.1, block_depth=5, sf2=0.1, ell=0.2, free_thresh=0.3, occupied_thresh=0.7, var_thresh=0.2,
               prior_A=0.001, prior_B=0.001, original_size=True, min_W=0.001)   # config/methods/bgklvoctomap.yaml
LV_STATS = ["n_hits", "n_rays", "n_samples", "n_bbox_blocks", "n_info_blocks", "voxels_visited", "voxel_updates", "rows",
            "t_frontend", "t_infer", "t_total"]


class OracleLVMap:
    """CPU BGKLVOctoMap restatement (constructor argument order of include/bgklvoctomap/bgklvoctomap.h)."""

    def __init__(self, resolution=0.1, block_depth=4, sf2=1.0, ell=1.0, free_thresh=0.3, occupied_thresh=0.7,
                 var_thresh=1.0, prior_A=1.0, prior_B=1.0, original_size=True, min_W=0.1, omp=False):
        self.L = lib(omp)  # omp: the OpenMP build (hits / blocks in parallel, the same values in the same order)
        self.h = self.L.orc_lv_map_create(resolution, block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh,
                                          prior_A, prior_B, int(original_size), min_W)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_lv_map_destroy(self.h)
            self.h = None

    def insert_pointcloud(self, xyz, origin, ds_resolution, free_res=2.0, max_range=-1.0):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        origin = np.ascontiguousarray(origin, np.float32)
        self.L.orc_lv_insert_pointcloud(self.h, xyz, xyz.shape[0], origin, ds_resolution, free_res, max_range)

    def stats(self):
        a = np.zeros(len(LV_STATS), np.float64)
        self.L.orc_lv_stats(self.h, a)
        return dict(zip(LV_STATS, a.tolist()))

    def training_data(self, xyz, origin, ds_resolution, free_res, max_range):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        origin = np.ascontiguousarray(origin, np.float32)
        nr = C.c_int64()
        n = self.L.orc_lv_training_data(self.h, xyz, xyz.shape[0], origin, ds_resolution, free_res, max_range, None, 0,
                                        None, 0, C.byref(nr))
        xy, rays = np.zeros((n, 4), np.float32), np.zeros((nr.value, 6), np.float32)
        self.L.orc_lv_training_data(self.h, xyz, xyz.shape[0], origin, ds_resolution, free_res, max_range,
                                    xy.ctypes.data, n, rays.ctypes.data, nr.value, C.byref(nr))
        return xy, rays

    def leaves(self, all_blocks=False):
        n = self.L.orc_lv_dump_leaves(self.h, int(all_blocks), None, None, None, None, None, None, None, None, 0)
        out = dict(block_key=np.zeros(n, np.int64), node_key=np.zeros(n, np.int64), loc=np.zeros((n, 3), np.float32),
                   size=np.zeros(n, np.float32), A=np.zeros(n, np.float32), B=np.zeros(n, np.float32),
                   state=np.zeros(n, np.uint8), classified=np.zeros(n, np.uint8))
        self.L.orc_lv_dump_leaves(self.h, int(all_blocks), *[out[k].ctypes.data for k in
                                  ("block_key", "node_key", "loc", "size", "A", "B", "state", "classified")], n)
        return out


def get_training_data(xyz, origin, ds_resolution, free_res, max_range):
    L = lib()
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    origin = np.ascontiguousarray(origin, np.float32)
    n = L.orc_get_training_data(xyz, xyz.shape[0], origin, ds_resolution, free_res, max_range, None, 0)
    out = np.zeros((n, 4), np.float32)
    L.orc_get_training_data(xyz, xyz.shape[0], origin, ds_resolution, free_res, max_range,
                            out.ctypes.data_as(C.c_void_p), n)
    return out


def voxel_grid(xyz, leaf):
    xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    out = np.zeros_like(xyz)
    n = lib().orc_voxel_grid(xyz, xyz.shape[0], leaf, out)
    return out[:n].copy()


def kernel(r, sf2=1.0):
    r = np.ascontiguousarray(r, np.float32)
    out = np.zeros_like(r)
    lib().orc_kernel_array(r, r.size, sf2, out)
    return out


def bgk_predict(sf2, ell, xs, x, y):
    xs = np.ascontiguousarray(xs, np.float32).reshape(-1, 3)
    x = np.ascontiguousarray(x, np.float32).reshape(-1, 3)
    y = np.ascontiguousarray(y, np.float32)
    ybar = np.zeros(xs.shape[0], np.float32)
    kbar = np.zeros(xs.shape[0], np.float32)
    lib().orc_bgk_predict(sf2, ell, xs, xs.shape[0], x, y, x.shape[0], ybar, kbar)
    return ybar, kbar


# ---------------------------------------------------------------------------
# compiled reference layer (oracle/_ref)
# ---------------------------------------------------------------------------
_ref = None
_ref_lv = None


def _bind_ref(R):
    """prototypes of the entry points both builds of oracle/ref_harness.cpp export"""
    R.ref_configure.argtypes = [C.c_float, C.c_int] + [C.c_float] * 7
    R.ref_block_size.restype = C.c_float
    R.ref_lut.restype = C.c_int
    R.ref_lut.argtypes = [C.c_int, C.c_int, f32p]
    R.ref_block_to_hash_key.restype = C.c_int64
    R.ref_block_to_hash_key.argtypes = [C.c_float] * 3
    R.ref_hash_key_to_block.argtypes = [C.c_int64, f32p]
    R.ref_get_extended_block.argtypes = [C.c_int64, i64p]
    R.ref_block_new.restype = C.c_void_p
    R.ref_block_new.argtypes = [C.c_float] * 3
    R.ref_block_free.argtypes = [C.c_void_p]
    R.ref_block_extended.argtypes = [C.c_void_p, i64p]
    R.ref_block_grid.argtypes = [C.c_void_p, f32p, i32p, C.POINTER(C.c_int32), f32p]
    R.ref_block_leaves.restype = C.c_int
    R.ref_block_leaves.argtypes = [C.c_void_p, i32p, f32p, f32p, C.c_int]
    R.ref_block_update.argtypes = [C.c_void_p, C.c_int32, C.c_float, C.c_float]
    R.ref_block_prune.restype = C.c_int
    R.ref_block_prune.argtypes = [C.c_void_p]
    R.ref_block_node.restype = C.c_int
    R.ref_block_node.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                 C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    R.ref_node_sequence.argtypes = [f32p, f32p, C.c_int, f32p, f32p, u8p, f32p, f32p]
    return R


def ref_lv():
    """ctypes handle of the reference's std-only BGK-LV layer (node, 28-bit key tree, block, point6f), or None."""
    global _ref_lv
    if _ref_lv is None:
        build()
        path = os.path.join(_HERE, "_ref", "libla3dm_ref_lv.so")
        if not os.path.exists(path):
            return None
        R = _bind_ref(C.CDLL(path))
        R.ref_configure_lv.argtypes = [C.c_int, C.c_float]
        R.ref_node_to_hash_key.restype = C.c_int32
        R.ref_node_to_hash_key.argtypes = [C.c_int, C.c_uint32]
        R.ref_hash_key_to_node.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
        R.ref_node_ctor.argtypes = [C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_uint8),
                                    C.POINTER(C.c_float), C.POINTER(C.c_float)]
        R.ref_point6f.argtypes = [f32p] * 6
        _ref_lv = R
    return _ref_lv


def ref_available():
    build()
    return os.path.exists(os.path.join(_HERE, "_ref", "libla3dm_ref.so"))


def ref():
    """ctypes handle of the reference's std-only layer, or None if not built."""
    global _ref
    if _ref is None:
        if not ref_available():
            return None
        R = _bind_ref(C.CDLL(os.path.join(_HERE, "_ref", "libla3dm_ref.so")))
        R.ref_rtree_new.restype = C.c_void_p
        R.ref_rtree_new.argtypes = [f32p, C.c_int]
        R.ref_rtree_free.argtypes = [C.c_void_p]
        R.ref_rtree_block_query.restype = C.c_int
        R.ref_rtree_block_query.argtypes = [C.c_void_p, C.c_int64, i32p, C.c_int]
        R.ref_rtree_box_query.restype = C.c_int
        R.ref_rtree_box_query.argtypes = [C.c_void_p, f32p, f32p, i32p, C.c_int]
        _ref = R
    return _ref


def set_sum_mode(mode=0, omp=False):
    """0 = the reference's fp32 summation order (default), 1 = double accumulators over all 7 neighbours, rounded once: the
    counterpart of the device option bgk_sum = 1 (oracle/la3dm_oracle.cpp orc_set_sum_mode).  Process-global."""
    lib(omp).orc_set_sum_mode(int(mode))


def set_gp_mode(mode=0, omp=False):
    """GP sensitivity switch of the restatement: 0 = FMA chains in ascending k (what every parity test uses), 1 = an
    emulation of Eigen 3.3.7's order of operations on x86-64 without FMA (oracle/la3dm_oracle.cpp, gp_train_eigen).
    Process-global per library build."""
    lib(omp).orc_set_gp_mode(int(mode))


def set_modes(trig=0, grid_sort=0, omp=False):
    """sensitivity switches of the restatement (oracle/la3dm_oracle.cpp): trig 1 = Eigen 3.3.7 SSE packet psin / pcos
    instead of correctly rounded sin / cos; grid_sort 1 = pcl::VoxelGrid's unstable std::sort on the cell index alone.
    Process-global per library build; (0, 0) is the default every parity test uses."""
    lib(omp).orc_set_modes(int(trig), int(grid_sort))
