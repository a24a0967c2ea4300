"""ctypes bindings of the two product libraries.

  libla3dm_hip.so   C ABI of the device hot path        (include/la3dm_hip.h)
  libla3dm_map.so   host-side BGKOctoMap, C view         (include/la3dm_map.h)

Both are built in-tree by la3dm_amd/csrc/Makefile.  There is no Python or CPU fallback:
if a library is missing or no HIP device is present, loading / map creation raises.
"""
import ctypes as C
import os

import numpy as np

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
HIP_SO = os.path.join(_CSRC, "libla3dm_hip.so")
MAP_SO = os.path.join(_CSRC, "libla3dm_map.so")

class GatherSeg(C.Structure):
    """la3dm_gather_seg (include/la3dm_hip.h)"""
    _fields_ = [("base", C.c_void_p), ("offset", C.POINTER(C.c_uint64)), ("bytes", C.POINTER(C.c_uint64))]


# la3dm_allgatherv_fn: int fn(void *user, const la3dm_gather_seg *segs, uint32_t nseg, uint32_t world, uint32_t rank, void *stream)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(GatherSeg), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p)

LEAF_UPDATED = 0x80
SCAN_UPDATE_UNGATED = 0x1


class Params(C.Structure):
    _fields_ = [("resolution", C.c_float), ("block_depth", C.c_int32), ("sf2", C.c_float), ("ell", C.c_float),
                ("free_thresh", C.c_float), ("occupied_thresh", C.c_float), ("var_thresh", C.c_float),
                ("prior_A", C.c_float), ("prior_B", C.c_float), ("device", C.c_int32),
                ("lut_xyz", C.c_void_p), ("lut_count", C.c_uint32), ("variant", C.c_int32), ("noise", C.c_float),
                ("l", C.c_float), ("min_ivar", C.c_float), ("max_ivar", C.c_float), ("min_known_ivar", C.c_float), ("min_W", C.c_float)]


class BgkScan(C.Structure):
    _fields_ = [("train_xyzy", C.c_void_p), ("train_off", C.c_void_p), ("n_train_pts", C.c_uint32),
                ("n_train_blk", C.c_uint32), ("nbr", C.c_void_p), ("blk_center", C.c_void_p),
                ("leaf_off", C.c_void_p), ("n_test_blk", C.c_uint32), ("n_leaf", C.c_uint32),
                ("leaf_key", C.c_void_p), ("alpha", C.c_void_p), ("beta", C.c_void_p), ("state", C.c_void_p),
                ("flags", C.c_uint32), ("train_max_n", C.c_uint32), ("train_sum_n2", C.c_uint64)]


class LvScan(C.Structure):
    _fields_ = [("samples", C.c_void_p), ("sorted", C.c_void_p), ("n_samples", C.c_uint32), ("rays", C.c_void_p),
                ("n_rays", C.c_uint32), ("cell_off", C.c_void_p), ("cell_min", C.c_int32 * 3), ("cell_dim", C.c_int32 * 3),
                ("n_blk", C.c_uint32), ("blk_center", C.c_void_p), ("blk_cell0", C.c_void_p), ("alpha", C.c_void_p),
                ("beta", C.c_void_p), ("state", C.c_void_p)]


class BgkCounters(C.Structure):
    _fields_ = [("n_tiles", C.c_uint64), ("scratch_bytes", C.c_uint64)]


class ScanStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_hits", "n_frees", "n_bbox_blocks", "n_train_blocks", "n_test_blocks",
                                           "voxel_updates", "train_reads", "pair_evals", "n_tiles")] + \
               [(n, C.c_double) for n in ("t_frontend", "t_partition", "t_pack", "t_device", "t_commit", "t_prune",
                                          "t_total", "t_gather")]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class DevmapStats(C.Structure):
    """la3dm_devmap_stats (include/la3dm_hip.h)"""
    _fields_ = [(k, C.c_uint64) for k in ("n_hits", "n_frees", "n_train_blocks", "n_test_blocks", "n_bbox_blocks",
                                          "voxel_updates", "train_reads", "pair_evals", "n_blocks")] + \
               [("n_passes", C.c_uint32)] + \
               [(k, C.c_double) for k in ("t_frontend", "t_partition", "t_pack", "t_kernel", "t_commit", "t_total", "t_gather")]


HIP_SYMBOLS = ["la3dm_device_count", "la3dm_version", "la3dm_create", "la3dm_destroy", "la3dm_last_error",
               "la3dm_set_option", "la3dm_get_option", "la3dm_bgk_scan_host", "la3dm_bgk_scan_device", "la3dm_gp_scan_host",
               "la3dm_gp_scan_device", "la3dm_bgklv_scan_host", "la3dm_bgklv_scan_device", "la3dm_kernel_times", "la3dm_diag_eval", "la3dm_diag_sweep",
               "la3dm_devmap_create", "la3dm_devmap_destroy", "la3dm_devmap_insert_pointcloud_host",
               "la3dm_devmap_insert_pointcloud_device", "la3dm_devmap_block_count", "la3dm_devmap_download",
               "la3dm_devmap_training_data", "la3dm_devmap_diag_add_repeat", "la3dm_bgkl_scan_host",
               "la3dm_bgkl_scan_device", "la3dm_diag_mfma_chain", "la3dm_devmap_search_host",
               "la3dm_devmap_export_cells", "la3dm_devmap_key_bounds",
               "la3dm_devmap_insert_training_data_host", "la3dm_devmap_set_shard", "la3dm_devmap_wait_event",
               "la3dm_devmap_lv_stats_get", "la3dm_devmap_lv_set_original_size", "la3dm_devmap_lv_training",
               "la3dm_devmap_diag_scan", "la3dm_devmap_diag_sort"]
MAP_SYMBOLS = ["la3dm_map_create", "la3dm_map_create_gp", "la3dm_map_create_lv", "la3dm_map_lv_training",
               "la3dm_map_lv_stats", "la3dm_map_lv_prepare", "la3dm_map_lv_packed", "la3dm_map_lv_commit", "la3dm_map_destroy", "la3dm_map_last_error", "la3dm_map_insert_pointcloud", "la3dm_map_insert_pointcloud_device",
               "la3dm_map_insert_training_data", "la3dm_map_prepare", "la3dm_map_prepare_training_data",
               "la3dm_map_packed", "la3dm_map_commit", "la3dm_map_ctx", "la3dm_map_stats", "la3dm_map_training_size",
               "la3dm_map_training_data", "la3dm_map_block_size", "la3dm_map_block_count", "la3dm_map_leaf_count",
               "la3dm_map_dump_leaves", "la3dm_map_search", "la3dm_map_get_bbox", "la3dm_map_block_to_hash_key",
               "la3dm_map_hash_key_to_block", "la3dm_map_extended_block", "la3dm_map_lut",
               "la3dm_map_set_device_resident", "la3dm_map_is_device_resident", "la3dm_map_raycast",
               "la3dm_map_block_grid", "la3dm_map_create_l", "la3dm_map_l_training",
               "la3dm_map_search_many", "la3dm_map_export_cells", "la3dm_map_set_shard", "la3dm_map_resolution",
               "la3dm_map_block_depth", "la3dm_map_set_resolution", "la3dm_map_set_block_depth"]

_hip = None
_map = None


def hip():
    """libla3dm_hip.so (raises OSError if it has not been built)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise OSError(f"{HIP_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no fallback path)")
        L = C.CDLL(HIP_SO, mode=C.RTLD_GLOBAL)
        L.la3dm_device_count.restype = C.c_int
        L.la3dm_version.restype = C.c_char_p
        L.la3dm_create.restype = C.c_int
        L.la3dm_create.argtypes = [C.POINTER(Params), C.POINTER(C.c_void_p)]
        L.la3dm_destroy.argtypes = [C.c_void_p]
        L.la3dm_last_error.restype = C.c_char_p
        L.la3dm_last_error.argtypes = [C.c_void_p]
        L.la3dm_set_option.restype = C.c_int
        L.la3dm_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.la3dm_get_option.restype = C.c_int
        L.la3dm_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int)]
        L.la3dm_bgk_scan_host.restype = C.c_int
        L.la3dm_bgk_scan_host.argtypes = [C.c_void_p, C.POINTER(BgkScan), C.POINTER(BgkCounters)]
        L.la3dm_bgk_scan_device.restype = C.c_int
        L.la3dm_bgk_scan_device.argtypes = [C.c_void_p, C.POINTER(BgkScan), C.c_void_p, C.POINTER(BgkCounters)]
        L.la3dm_diag_mfma_chain.restype = C.c_int
        L.la3dm_diag_mfma_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]
        L.la3dm_bgkl_scan_host.restype = C.c_int
        L.la3dm_bgkl_scan_host.argtypes = [C.c_void_p, C.POINTER(BgkScan), C.POINTER(BgkCounters)]
        L.la3dm_bgkl_scan_device.restype = C.c_int
        L.la3dm_bgkl_scan_device.argtypes = [C.c_void_p, C.POINTER(BgkScan), C.c_void_p, C.POINTER(BgkCounters)]
        L.la3dm_gp_scan_host.restype = C.c_int
        L.la3dm_gp_scan_host.argtypes = [C.c_void_p, C.POINTER(BgkScan), C.POINTER(BgkCounters)]
        L.la3dm_gp_scan_device.restype = C.c_int
        L.la3dm_gp_scan_device.argtypes = [C.c_void_p, C.POINTER(BgkScan), C.c_void_p, C.POINTER(BgkCounters)]
        L.la3dm_bgklv_scan_host.restype = C.c_int
        L.la3dm_bgklv_scan_host.argtypes = [C.c_void_p, C.POINTER(LvScan), C.POINTER(BgkCounters)]
        L.la3dm_bgklv_scan_device.restype = C.c_int
        L.la3dm_bgklv_scan_device.argtypes = [C.c_void_p, C.POINTER(LvScan), C.c_void_p, C.POINTER(BgkCounters)]
        L.la3dm_kernel_times.restype = C.c_int
        L.la3dm_kernel_times.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        L.la3dm_diag_sweep.restype = C.c_int
        L.la3dm_diag_sweep.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
        L.la3dm_diag_eval.restype = C.c_int
        L.la3dm_diag_eval.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
        L.la3dm_devmap_create.restype = C.c_int
        L.la3dm_devmap_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.la3dm_devmap_destroy.argtypes = [C.c_void_p]
        L.la3dm_devmap_insert_pointcloud_host.restype = C.c_int
        L.la3dm_devmap_insert_pointcloud_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                                          C.c_float, C.c_float, C.c_float, C.POINTER(DevmapStats)]
        L.la3dm_devmap_insert_pointcloud_device.restype = C.c_int
        L.la3dm_devmap_insert_pointcloud_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_float,
                                                            C.c_float, C.c_float, C.POINTER(DevmapStats)]
        L.la3dm_devmap_block_count.restype = C.c_int
        L.la3dm_devmap_block_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.la3dm_devmap_download.restype = C.c_int
        L.la3dm_devmap_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.la3dm_devmap_diag_add_repeat.restype = C.c_int
        L.la3dm_devmap_diag_scan.restype = C.c_int
        L.la3dm_devmap_diag_scan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.la3dm_devmap_diag_sort.restype = C.c_int
        L.la3dm_devmap_diag_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.la3dm_devmap_diag_add_repeat.argtypes = [C.c_void_p] + [C.c_void_p] * 3 + [C.c_uint32] + [C.c_void_p] * 2
        L.la3dm_devmap_search_host.restype = C.c_int
        L.la3dm_devmap_search_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 4
        L.la3dm_devmap_training_data.restype = C.c_int
        L.la3dm_devmap_training_data.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
        _hip = L
    return _hip


def maplib():
    global _map
    if _map is None:
        hip()
        if not os.path.exists(MAP_SO):
            raise OSError(f"{MAP_SO} is missing: build it first (no fallback path)")
        M = C.CDLL(MAP_SO)
        f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        M.la3dm_map_create.restype = C.c_void_p
        M.la3dm_map_create.argtypes = [C.c_float, C.c_int] + [C.c_float] * 7 + [C.c_int]
        M.la3dm_map_create_l.restype = C.c_void_p
        M.la3dm_map_create_l.argtypes = [C.c_float, C.c_int] + [C.c_float] * 7 + [C.c_int]
        M.la3dm_map_l_training.restype = C.c_uint64
        M.la3dm_map_l_training.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        M.la3dm_map_create_gp.restype = C.c_void_p
        M.la3dm_map_create_gp.argtypes = [C.c_float, C.c_int] + [C.c_float] * 9 + [C.c_int]
        M.la3dm_map_create_lv.restype = C.c_void_p
        M.la3dm_map_create_lv.argtypes = [C.c_float, C.c_int] + [C.c_float] * 7 + [C.c_int, C.c_float, C.c_int]
        M.la3dm_map_lv_training.restype = C.c_uint64
        M.la3dm_map_lv_training.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        M.la3dm_map_lv_stats.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.float64)]
        M.la3dm_map_lv_prepare.restype = C.c_int
        M.la3dm_map_lv_prepare.argtypes = [C.c_void_p, f32p, C.c_uint64, f32p, C.c_float, C.c_float, C.c_float]
        M.la3dm_map_lv_packed.restype = C.c_int
        M.la3dm_map_lv_packed.argtypes = [C.c_void_p, C.POINTER(LvScan)]
        M.la3dm_map_lv_commit.restype = C.c_int
        M.la3dm_map_lv_commit.argtypes = [C.c_void_p]
        M.la3dm_map_destroy.argtypes = [C.c_void_p]
        M.la3dm_map_block_grid.argtypes = [C.c_void_p, f32p, f32p, np.ctypeslib.ndpointer(np.int32), C.POINTER(C.c_int32), f32p]
        M.la3dm_map_search_many.restype = C.c_int
        M.la3dm_map_search_many.argtypes = [C.c_void_p, f32p, C.c_uint64] + [C.c_void_p] * 4
        M.la3dm_map_export_cells.restype = C.c_int
        M.la3dm_map_export_cells.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        M.la3dm_map_raycast.restype = C.c_uint64
        M.la3dm_map_raycast.argtypes = [C.c_void_p, f32p, f32p] + [C.c_void_p] * 7 + [C.c_uint64]
        M.la3dm_map_set_shard.restype = C.c_int
        M.la3dm_map_set_shard.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, ALLGATHER_FN, C.c_void_p]
        M.la3dm_map_set_device_resident.restype = C.c_int
        M.la3dm_map_set_device_resident.argtypes = [C.c_void_p, C.c_int]
        M.la3dm_map_is_device_resident.restype = C.c_int
        M.la3dm_map_is_device_resident.argtypes = [C.c_void_p]
        M.la3dm_map_last_error.restype = C.c_char_p
        M.la3dm_map_insert_pointcloud_device.restype = C.c_int
        M.la3dm_map_insert_pointcloud_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, f32p, C.c_float, C.c_float, C.c_float]
        M.la3dm_map_insert_pointcloud.restype = C.c_int
        M.la3dm_map_insert_pointcloud.argtypes = [C.c_void_p, f32p, C.c_uint64, f32p, C.c_float, C.c_float, C.c_float]
        M.la3dm_map_insert_training_data.restype = C.c_int
        M.la3dm_map_insert_training_data.argtypes = [C.c_void_p, f32p, C.c_uint64]
        M.la3dm_map_prepare.restype = C.c_int
        M.la3dm_map_prepare.argtypes = [C.c_void_p, f32p, C.c_uint64, f32p, C.c_float, C.c_float, C.c_float]
        M.la3dm_map_prepare_training_data.restype = C.c_int
        M.la3dm_map_prepare_training_data.argtypes = [C.c_void_p, f32p, C.c_uint64, C.c_int]
        M.la3dm_map_packed.restype = C.c_int
        M.la3dm_map_packed.argtypes = [C.c_void_p, C.POINTER(BgkScan)]
        M.la3dm_map_commit.restype = C.c_int
        M.la3dm_map_commit.argtypes = [C.c_void_p]
        M.la3dm_map_ctx.restype = C.c_void_p
        M.la3dm_map_ctx.argtypes = [C.c_void_p]
        M.la3dm_map_stats.argtypes = [C.c_void_p, C.POINTER(ScanStats)]
        M.la3dm_map_training_size.restype = C.c_uint64
        M.la3dm_map_training_size.argtypes = [C.c_void_p]
        M.la3dm_map_training_data.argtypes = [C.c_void_p, f32p, C.c_uint64]
        M.la3dm_map_block_size.restype = C.c_float
        M.la3dm_map_block_size.argtypes = [C.c_void_p]
        M.la3dm_map_resolution.restype = C.c_float
        M.la3dm_map_resolution.argtypes = [C.c_void_p]
        M.la3dm_map_block_depth.argtypes = [C.c_void_p]
        M.la3dm_map_set_resolution.argtypes = [C.c_void_p, C.c_float]
        M.la3dm_map_set_block_depth.argtypes = [C.c_void_p, C.c_int]
        M.la3dm_map_block_count.restype = C.c_uint64
        M.la3dm_map_block_count.argtypes = [C.c_void_p]
        M.la3dm_map_leaf_count.restype = C.c_uint64
        M.la3dm_map_leaf_count.argtypes = [C.c_void_p]
        M.la3dm_map_dump_leaves.restype = C.c_uint64
        M.la3dm_map_dump_leaves.argtypes = [C.c_void_p] + [C.c_void_p] * 8 + [C.c_uint64]
        M.la3dm_map_search.restype = C.c_int
        M.la3dm_map_search.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float),
                                       C.POINTER(C.c_float), C.POINTER(C.c_uint8)]
        M.la3dm_map_get_bbox.argtypes = [C.c_void_p, f32p, f32p]
        M.la3dm_map_block_to_hash_key.restype = C.c_int64
        M.la3dm_map_block_to_hash_key.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        M.la3dm_map_hash_key_to_block.argtypes = [C.c_void_p, C.c_int64, f32p]
        M.la3dm_map_extended_block.argtypes = [C.c_void_p, C.c_int64, np.ctypeslib.ndpointer(np.int64)]
        M.la3dm_map_lut.restype = C.c_uint32
        M.la3dm_map_lut.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        _map = M
    return _map
