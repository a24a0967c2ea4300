"""Python mirror of la3dm::BGKOctoMap (reference include/bgkoctomap/bgkoctomap.h:26-367).

Same constructor argument order, same method names and argument meaning as the C++
class; all work is done by libla3dm_map.so (host bookkeeping, C++) and
libla3dm_hip.so (HIP kernels).  No Python compute path exists.
"""
import ctypes as C

import numpy as np

from . import _lib

FREE, OCCUPIED, UNKNOWN, PRUNED = 0, 1, 2, 3


class PackedScan:
    """Host view of one prepared scan (numpy arrays aliasing the map's buffers)."""

    def __init__(self, s: _lib.BgkScan):
        self.c = s

        def arr(ptr, n, dt):
            if not ptr or n == 0:
                return np.zeros(0, dt)
            buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dt, count=n)

        self.train_xyzy = arr(s.train_xyzy, 4 * s.n_train_pts, np.float32).reshape(-1, 4)
        self.train_off = arr(s.train_off, s.n_train_blk + 1, np.uint32)
        self.nbr = arr(s.nbr, 7 * s.n_test_blk, np.int32).reshape(-1, 7)
        self.blk_center = arr(s.blk_center, 3 * s.n_test_blk, np.float32).reshape(-1, 3)
        self.leaf_off = arr(s.leaf_off, s.n_test_blk + 1, np.uint32)
        self.leaf_key = arr(s.leaf_key, s.n_leaf, np.uint32)
        self.alpha = arr(s.alpha, s.n_leaf, np.float32)
        self.beta = arr(s.beta, s.n_leaf, np.float32)
        self.state = arr(s.state, s.n_leaf, np.uint8)
        self.flags = s.flags
        self.n_test_blk = s.n_test_blk
        self.n_leaf = s.n_leaf
        self.n_train_pts = s.n_train_pts
        self.n_train_blk = s.n_train_blk


class BGKOctoMap:
    def __init__(self, resolution=0.1, block_depth=4, sf2=1.0, ell=1.0, free_thresh=0.3, occupied_thresh=0.7,
                 var_thresh=1.0, prior_A=1.0, prior_B=1.0, device=0):
        self._M = _lib.maplib()
        self._h = self._M.la3dm_map_create(resolution, block_depth, sf2, ell, free_thresh, occupied_thresh,
                                           var_thresh, prior_A, prior_B, device)
        if not self._h:
            raise RuntimeError(self._M.la3dm_map_last_error().decode())
        self.resolution = resolution
        self.block_depth = block_depth
        self.device = device

    _gp = False

    def set_device_resident(self, on=True):
        """Device-resident mode: the block pool lives in HBM and insert_pointcloud runs start to finish on the
        GPU (front end, partition, predict + fuse, write-back, prune); queries refresh a host mirror lazily."""
        self._chk(self._M.la3dm_map_set_device_resident(self._h, 1 if on else 0))
        return self

    def is_device_resident(self):
        return bool(self._M.la3dm_map_is_device_resident(self._h))

    def set_shard(self, rank, world, allgatherv):
        """Block-sharded insert_pointcloud over `world` replicas of this map, one process per GPU (every process
        inserts the same clouds): rank r predicts + fuses its contiguous range of the test blocks, then
        `allgatherv(segments, world, rank, stream)` is called once per pass — segments = [(base_ptr, offsets, nbytes), ...]
        (four: the scan's alpha, beta, state and leaf-key arrays; rank q owns bytes [offsets[q], offsets[q] + nbytes[q]) of each)
        — and must queue, on the HIP stream `stream` (an integer handle) or ordered against it, an in-place
        all-gather-v of every segment; nothing synchronises the host (la3dm_devmap_set_shard in include/la3dm_hip.h;
        la3dm_amd.sharding.torch_allgather builds one on torch.distributed).  world = 1 switches it off."""
        self._shard_error = None

        def _cb(user, segs, nseg, nranks, r, stream):
            try:
                seg_list = [(int(segs[i].base), [int(segs[i].offset[q]) for q in range(nranks)],
                             [int(segs[i].bytes[q]) for q in range(nranks)]) for i in range(nseg)]
                allgatherv(seg_list, int(nranks), int(r), int(stream or 0))
                return 0
            except Exception as e:           # never let an exception cross the C boundary
                self._shard_error = e
                return 1
        self._shard_cb = _lib.ALLGATHER_FN(_cb) if world > 1 else _lib.ALLGATHER_FN(0)
        self._chk(self._M.la3dm_map_set_shard(self._h, rank, world, self._shard_cb, None))
        return self

    def __del__(self):
        if getattr(self, "_h", None):
            self._M.la3dm_map_destroy(self._h)
            self._h = None

    def _chk(self, rc):
        if rc < 0:
            err, self._shard_error = getattr(self, "_shard_error", None), None
            if err is not None:          # the all-gather callback raised: that is the cause
                raise RuntimeError(self._M.la3dm_map_last_error().decode()) from err
            raise RuntimeError(self._M.la3dm_map_last_error().decode())
        return rc

    # -- reference API ------------------------------------------------------
    def get_resolution(self):
        return self._M.la3dm_map_resolution(self._h)

    def get_block_depth(self):
        return self._M.la3dm_map_block_depth(self._h)

    def set_resolution(self, resolution):
        """BGKOctoMap::set_resolution (reference src/bgkoctomap/bgkoctomap.cpp:66-72): empty maps only (RuntimeError otherwise)"""
        self._chk(self._M.la3dm_map_set_resolution(self._h, float(resolution)))
        self.resolution = self.get_resolution()
        return self

    def set_block_depth(self, block_depth):
        """BGKOctoMap::set_block_depth (reference src/bgkoctomap/bgkoctomap.cpp:74-80): empty maps only (RuntimeError otherwise)"""
        self._chk(self._M.la3dm_map_set_block_depth(self._h, int(block_depth)))
        self.block_depth = self.get_block_depth()
        return self

    def get_block_size(self):
        return self._M.la3dm_map_block_size(self._h)

    def insert_pointcloud(self, cloud, origin, ds_resolution, free_res=2.0, max_range=-1.0):
        """BGKOctoMap::insert_pointcloud(cloud, origin, ds_resolution, free_res, max_range)
        (include/bgkoctomap/bgkoctomap.h:82-84, src/bgkoctomap/bgkoctomap.cpp:214-366): ds_resolution < 0 skips the
        voxel-grid filter, max_range <= 0 the range gate; an empty training set is a silent no-op."""
        xyz = np.ascontiguousarray(cloud, np.float32).reshape(-1, 3)
        o = np.ascontiguousarray(origin, np.float32)
        self._chk(self._M.la3dm_map_insert_pointcloud(self._h, xyz, xyz.shape[0], o, ds_resolution, free_res,
                                                      max_range))

    def insert_pointcloud_device(self, d_xyz, n, origin, ds_resolution, free_res=2.0, max_range=-1.0):
        """insert_pointcloud for a cloud already in HBM: d_xyz = device address of n packed float32 xyz triples on the
        map's GPU (e.g. torch_tensor.data_ptr() of a contiguous (n, 3) float32 CUDA tensor); device-resident maps only.
        The insert runs on the library's own stream: the cloud must be complete before the call — synchronise the stream
        that produced it (torch.cuda.current_stream().synchronize()) or order the insert behind a hipEvent with
        la3dm_devmap_wait_event (include/la3dm_hip.h)."""
        o = np.ascontiguousarray(origin, np.float32)
        self._chk(self._M.la3dm_map_insert_pointcloud_device(self._h, int(d_xyz), int(n), o, ds_resolution, free_res, max_range))
        return self

    def insert_training_data(self, xyzy):
        """BGKOctoMap::insert_training_data(GPPointCloud) (bgkoctomap.h:86, bgkoctomap.cpp:82-212): rows x, y, z, label;
        every leaf of every test block is updated (no kbar gate)."""
        a = np.ascontiguousarray(xyzy, np.float32).reshape(-1, 4)
        self._chk(self._M.la3dm_map_insert_training_data(self._h, a, a.shape[0]))

    def get_bbox(self):
        """BGKOctoMap::get_bbox (bgkoctomap.h:89, bgkoctomap.cpp:368-381): block centres +- half a block"""
        lo, hi = np.zeros(3, np.float32), np.zeros(3, np.float32)
        self._M.la3dm_map_get_bbox(self._h, lo, hi)
        return lo, hi

    def search(self, x, y, z):
        """BGKOctoMap::search(x, y, z) (bgkoctomap.h:315-319, bgkoctomap.cpp:554-567) -> (exists, alpha, beta, state)"""
        a, b, s = C.c_float(), C.c_float(), C.c_uint8()
        e = self._M.la3dm_map_search(self._h, x, y, z, C.byref(a), C.byref(b), C.byref(s))
        return bool(e), a.value, b.value, s.value

    def search_many(self, points):
        """search(x, y, z) for an (n, 3) array of points -> dict(exists, A, B, state); a device-resident map answers
        from the device pool (no host mirror refresh)"""
        q = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        n = q.shape[0]
        out = dict(exists=np.zeros(n, np.uint8), A=np.zeros(n, np.float32), B=np.zeros(n, np.float32),
                   state=np.zeros(n, np.uint8))
        self._chk(self._M.la3dm_map_search_many(self._h, q, n, *[out[k].ctypes.data for k in ("exists", "A", "B", "state")]))
        return out

    def raycast(self, start, end, cap=4096):
        """RayCaster(map, start, end) driven to its end: dict of p, block_key, node_key, valid, A, B, state per step."""
        s3, e3 = np.ascontiguousarray(start, np.float32), np.ascontiguousarray(end, np.float32)
        out = dict(p=np.zeros((cap, 3), np.float32), block_key=np.zeros(cap, np.int64), node_key=np.zeros(cap, np.int32),
                   valid=np.zeros(cap, np.uint8), A=np.zeros(cap, np.float32), B=np.zeros(cap, np.float32),
                   state=np.zeros(cap, np.uint8))
        n = self._M.la3dm_map_raycast(self._h, s3, e3, *[out[k].ctypes.data for k in
                                                          ("p", "block_key", "node_key", "valid", "A", "B", "state")], cap)
        n = min(int(n), cap)
        return {k: v[:n] for k, v in out.items()}

    def export_cells(self, state="occupied", original_size=True, min_z=0.0, max_z=0.0):
        """Cube lists of the map (the static node's publish loop + MarkerArrayPub::insert_point3d / heightMapColor,
        reference bgkoctomap_static_node.cpp:101-136, markerarray_pub.h:21-147, minus ROS): dict of cells (n, 4)
        {x, y, z, size}, rgba (n, 4), level (n,) = (int) log2(size / resolution).  state "occupied": coloured by
        height between min_z and max_z (equal: the map's bbox); "free": coloured by probability.  original_size
        False expands collapsed leaves (get_pruned_locs).  Device-resident maps are scanned on the GPU."""
        st = {"occupied": 1, "free": 0}[state]
        n = C.c_uint64(0)
        if self._M.la3dm_map_export_cells(self._h, st, int(bool(original_size)), min_z, max_z, None, None, None, 0, C.byref(n)) != 0:
            raise RuntimeError(self._M.la3dm_map_last_error().decode())
        k = int(n.value)
        out = dict(cells=np.zeros((k, 4), np.float32), rgba=np.zeros((k, 4), np.float32), level=np.zeros(k, np.int32))
        if k and self._M.la3dm_map_export_cells(self._h, st, int(bool(original_size)), min_z, max_z, out["cells"].ctypes.data,
                                                out["rgba"].ctypes.data, out["level"].ctypes.data, k, C.byref(n)) != 0:
            raise RuntimeError(self._M.la3dm_map_last_error().decode())
        return out

    def leaves(self):
        """All leaves (begin_leaf()..end_leaf()), blocks by ascending hash key, leaves in
        LeafIterator order: dict of block_key, node_key, loc, size, A, B, state, classified."""
        n = self._M.la3dm_map_leaf_count(self._h)
        out = dict(block_key=np.zeros(n, np.int64), node_key=np.zeros(n, np.int32), loc=np.zeros((n, 3), np.float32),
                   size=np.zeros(n, np.float32), A=np.zeros(n, np.float32), B=np.zeros(n, np.float32),
                   state=np.zeros(n, np.uint8), classified=np.zeros(n, np.uint8))
        m = self._M.la3dm_map_dump_leaves(self._h, *[out[k].ctypes.data for k in
                                                     ("block_key", "node_key", "loc", "size", "A", "B", "state",
                                                      "classified")], n)
        assert m <= n
        return {k: v[:m] for k, v in out.items()} if m < n else out   # (an LV map only reports touched blocks)

    def block_count(self):
        return self._M.la3dm_map_block_count(self._h)

    # -- split form ---------------------------------------------------------
    def prepare(self, cloud, origin, ds_resolution, free_res=2.0, max_range=-1.0):
        xyz = np.ascontiguousarray(cloud, np.float32).reshape(-1, 3)
        o = np.ascontiguousarray(origin, np.float32)
        return bool(self._chk(self._M.la3dm_map_prepare(self._h, xyz, xyz.shape[0], o, ds_resolution, free_res,
                                                        max_range)))

    def prepare_training_data(self, xyzy, ungated=False):
        a = np.ascontiguousarray(xyzy, np.float32).reshape(-1, 4)
        return bool(self._chk(self._M.la3dm_map_prepare_training_data(self._h, a, a.shape[0], int(ungated))))

    def packed(self):
        s = _lib.BgkScan()
        self._chk(self._M.la3dm_map_packed(self._h, C.byref(s)))
        return PackedScan(s)

    def commit(self):
        self._chk(self._M.la3dm_map_commit(self._h))

    def ctx(self):
        return self._M.la3dm_map_ctx(self._h)

    def set_option(self, name, value):
        rc = _lib.hip().la3dm_set_option(self.ctx(), name.encode(), int(value))
        if rc != 0:
            raise RuntimeError(_lib.hip().la3dm_last_error(self.ctx()).decode())

    def get_option(self, name):
        v = C.c_int()
        if _lib.hip().la3dm_get_option(self.ctx(), name.encode(), C.byref(v)) != 0:
            raise RuntimeError(f"la3dm_get_option: unknown option {name}")
        return v.value

    def scan_host(self, packed: PackedScan):
        cnt = _lib.BgkCounters()
        fn = _lib.hip().la3dm_gp_scan_host if self._gp else _lib.hip().la3dm_bgk_scan_host
        rc = fn(self.ctx(), C.byref(packed.c), C.byref(cnt))
        if rc != 0:
            raise RuntimeError(_lib.hip().la3dm_last_error(self.ctx()).decode())
        return cnt

    def stats(self):
        s = _lib.ScanStats()
        self._M.la3dm_map_stats(self._h, C.byref(s))
        return s.as_dict()

    def training_data(self):
        n = self._M.la3dm_map_training_size(self._h)
        a = np.zeros((n, 4), np.float32)
        if n:
            self._M.la3dm_map_training_data(self._h, a, n)
        return a

    def diag_eval(self, op, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        rc = _lib.hip().la3dm_diag_eval(self.ctx(), op, x.ctypes.data, x.size, y.ctypes.data)
        if rc != 0:
            raise RuntimeError(_lib.hip().la3dm_last_error(self.ctx()).decode())
        return y

    def diag_sweep(self, what, lo, hi):
        """count fp32 inputs in [lo, hi] (floats, same sign) where a kernel shortcut differs from IEEE"""
        lo_b = int(np.float32(lo).view(np.uint32))
        hi_b = int(np.float32(hi).view(np.uint32))
        if lo_b > hi_b:
            lo_b, hi_b = hi_b, lo_b
        out = C.c_uint64()
        rc = _lib.hip().la3dm_diag_sweep(self.ctx(), what, lo_b, hi_b, C.byref(out))
        if rc != 0:
            raise RuntimeError(_lib.hip().la3dm_last_error(self.ctx()).decode())
        return out.value

    # host bookkeeping primitives
    def block_to_hash_key(self, x, y, z):
        return self._M.la3dm_map_block_to_hash_key(self._h, x, y, z)

    def hash_key_to_block(self, key):
        o = np.zeros(3, np.float32)
        self._M.la3dm_map_hash_key_to_block(self._h, key, o)
        return o

    def get_extended_block(self, key):
        o = np.zeros(7, np.int64)
        self._M.la3dm_map_extended_block(self._h, key, o)
        return o

    def lut(self):
        n = self._M.la3dm_map_lut(self._h, None, 0)
        a = np.zeros((n, 3), np.float32)
        self._M.la3dm_map_lut(self._h, a.ctypes.data, n)
        return a


class GPOctoMap(BGKOctoMap):
    """Python mirror of la3dm::GPOctoMap (reference include/gpoctomap/gpoctomap.h): GP regression per block
    (Matern-3/2, Cholesky) + BCM fusion.  leaves()["A"/"B"] are the nodes' m_ivar / ivar."""
    _gp = True

    def __init__(self, resolution=0.1, block_depth=4, sf2=1.0, ell=1.0, noise=0.01, l=100.0, min_var=0.001,
                 max_var=1000.0, max_known_var=0.02, free_thresh=0.3, occupied_thresh=0.7, device=0):
        self._M = _lib.maplib()
        self._h = self._M.la3dm_map_create_gp(resolution, block_depth, sf2, ell, noise, l, min_var, max_var,
                                              max_known_var, free_thresh, occupied_thresh, device)
        if not self._h:
            raise RuntimeError(self._M.la3dm_map_last_error().decode())
        self.resolution = resolution
        self.block_depth = block_depth
        self.device = device


class BGKLOctoMap(BGKOctoMap):
    """Python mirror of la3dm::BGKLOctoMap (reference include/bgkloctomap/bgkloctomap.h): block-level BGK whose
    free-space evidence are the beams themselves (line segments)."""

    def __init__(self, resolution=0.1, block_depth=4, sf2=1.0, ell=1.0, free_thresh=0.3, occupied_thresh=0.7,
                 var_thresh=1.0, prior_A=1.0, prior_B=1.0, device=0):
        self._M = _lib.maplib()
        self._h = self._M.la3dm_map_create_l(resolution, block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh,
                                             prior_A, prior_B, device)
        if not self._h:
            raise RuntimeError(self._M.la3dm_map_last_error().decode())
        self.resolution = resolution
        self.block_depth = block_depth
        self.device = device

    def l_training(self):
        """(samples (n, 4) x,y,z,beam index or -1; beams (m, 6)) of the last scan"""
        nr = C.c_uint64()
        n = self._M.la3dm_map_l_training(self._h, None, 0, None, 0, C.byref(nr))
        idx, rays = np.zeros(n, np.int32), np.zeros((nr.value, 6), np.float32)
        self._M.la3dm_map_l_training(self._h, idx.ctypes.data, n, rays.ctypes.data, nr.value, C.byref(nr))
        xy = self.training_data()
        xy[:, 3] = idx
        return xy, rays


LV_STATS = ["n_hits", "n_rays", "n_samples", "n_bbox_blocks", "n_packed_blocks", "n_info_blocks", "voxels",
            "voxel_updates", "t_frontend", "t_partition", "t_device", "t_commit", "t_total"]


class BGKLVOctoMap(BGKOctoMap):
    """Python mirror of la3dm::BGKLVOctoMap (reference include/bgklvoctomap/bgklvoctomap.h): hits + free-space line
    segments, per-voxel inference, variance-aware node (UNCERTAIN = 3, PRUNED = 4 in leaves()["state"])."""

    def __init__(self, resolution=0.1, block_depth=4, sf2=1.0, ell=1.0, free_thresh=0.3, occupied_thresh=0.7,
                 var_thresh=1.0, prior_A=1.0, prior_B=1.0, original_size=True, min_W=0.1, device=0):
        self._M = _lib.maplib()
        self._h = self._M.la3dm_map_create_lv(resolution, block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh,
                                              prior_A, prior_B, int(original_size), min_W, device)
        if not self._h:
            raise RuntimeError(self._M.la3dm_map_last_error().decode())
        self.resolution = resolution
        self.block_depth = block_depth
        self.device = device

    def lv_stats(self):
        a = np.zeros(13, np.float64)
        self._M.la3dm_map_lv_stats(self._h, a)
        return dict(zip(LV_STATS, a.tolist()))

    def lv_training(self):
        nr = C.c_uint64()
        n = self._M.la3dm_map_lv_training(self._h, None, 0, None, 0, C.byref(nr))
        s, r = np.zeros((n, 4), np.float32), np.zeros((nr.value, 6), np.float32)
        self._M.la3dm_map_lv_training(self._h, s.ctypes.data, n, r.ctypes.data, nr.value, C.byref(nr))
        return s, r

    def lv_prepare(self, cloud, origin, ds_resolution, free_res=2.0, max_range=-1.0):
        xyz = np.ascontiguousarray(cloud, np.float32).reshape(-1, 3)
        o = np.ascontiguousarray(origin, np.float32)
        return bool(self._chk(self._M.la3dm_map_lv_prepare(self._h, xyz, xyz.shape[0], o, ds_resolution, free_res,
                                                           max_range)))

    def lv_packed(self):
        s = _lib.LvScan()
        self._chk(self._M.la3dm_map_lv_packed(self._h, C.byref(s)))
        return s

    def lv_commit(self):
        self._chk(self._M.la3dm_map_lv_commit(self._h))
