/* la3dm_map.h — C binding of the host-side BGKOctoMap (la3dm_amd/csrc/host/bgkoctomap.h).
 *
 * The C++ class is the drop-in for the reference's la3dm::BGKOctoMap
 * (include/bgkoctomap/bgkoctomap.h:26-367); this flat C view exists for language
 * bindings (the Python tests and bench use it through ctypes).  Every function
 * returns 0 on success, negative on failure (la3dm_map_last_error()).
 */
#ifndef LA3DM_MAP_H
#define LA3DM_MAP_H
#include <stdint.h>
#include "la3dm_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct la3dm_map la3dm_map;

/* mirrors ScanStats */
typedef struct la3dm_scan_stats {
    uint64_t n_hits, n_frees, n_bbox_blocks, n_train_blocks, n_test_blocks;
    uint64_t voxel_updates, train_reads, pair_evals, n_tiles;
    double t_frontend, t_partition, t_pack, t_device, t_commit, t_prune, t_total;
    double t_gather; /* sharded device-resident insert with LA3DM_TIMING=1: the all-gather-v (otherwise part of t_device) */
} la3dm_scan_stats;

/* BGKOctoMap(resolution, block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh, prior_A, prior_B) */
la3dm_map *la3dm_map_create(float resolution, int block_depth, float sf2, float ell, float free_thresh,
                            float occupied_thresh, float var_thresh, float prior_A, float prior_B, int device);
/* GPOctoMap(resolution, block_depth, sf2, ell, noise, l, min_var, max_var, max_known_var, free_thresh, occupied_thresh)
 * (src/gpoctomap/gpoctomap.cpp:23-25); every other call below works on either map kind.  For a GP map the
 * leaf arrays A/B hold the node's m_ivar/ivar. */
la3dm_map *la3dm_map_create_gp(float resolution, int block_depth, float sf2, float ell, float noise, float l,
                               float min_var, float max_var, float max_known_var, float free_thresh,
                               float occupied_thresh, int device);
/* BGKLOctoMap(resolution, block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh, prior_A, prior_B)
 * (src/bgkloctomap/bgkloctomap.cpp:31-57): block-level BGK whose free-space evidence are beam segments. */
la3dm_map *la3dm_map_create_l(float resolution, int block_depth, float sf2, float ell, float free_thresh,
                              float occupied_thresh, float var_thresh, float prior_A, float prior_B, int device);
/* BGKLOctoMap only: beam index per training sample (-1 = hit) and beams (6 floats) of the last scan; returns the number
 * of samples (the samples themselves come from la3dm_map_training_data) */
uint64_t la3dm_map_l_training(const la3dm_map *m, int32_t *ray_idx, uint64_t cap, float *rays6, uint64_t cap_rays,
                              uint64_t *n_rays);
/* BGKLVOctoMap(resolution, block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh, prior_A, prior_B,
 * original_size, min_W) (src/bgklvoctomap/bgklvoctomap.cpp:33-43).  For an LV map la3dm_map_dump_leaves reports the
 * reference's LV state codes (UNCERTAIN = 3, PRUNED = 4), node_key = (depth << 28) + index, and only blocks that hold a
 * classified or collapsed leaf (the LV map allocates every block of each scan's bounding box). */
la3dm_map *la3dm_map_create_lv(float resolution, int block_depth, float sf2, float ell, float free_thresh,
                               float occupied_thresh, float var_thresh, float prior_A, float prior_B, int original_size,
                               float min_W, int device);
/* LV only: samples (x, y, z, ray) and segments (6 floats) of the last scan; counts via NULL pointers */
uint64_t la3dm_map_lv_training(const la3dm_map *m, float *samples4, uint64_t cap_samples, float *rays6, uint64_t cap_rays,
                               uint64_t *n_rays);
/* LV only: n_hits, n_rays, n_samples, n_bbox_blocks, n_packed_blocks, n_info_blocks, voxels, voxel_updates,
 * t_frontend, t_partition, t_device, t_commit, t_total */
int la3dm_map_lv_stats(const la3dm_map *m, double *out13);
/* LV only, split form: prepare -> la3dm_bgklv_scan_* on the packed arguments -> commit */
int la3dm_map_lv_prepare(la3dm_map *m, const float *xyz, uint64_t n, const float *origin3, float ds_resolution,
                         float free_res, float max_range);
int la3dm_map_lv_packed(la3dm_map *m, la3dm_lv_scan *out);
int la3dm_map_lv_commit(la3dm_map *m);
void la3dm_map_destroy(la3dm_map *m);
const char *la3dm_map_last_error(void);

/* insert_pointcloud(cloud, origin, ds_resolution, free_res, max_range); xyz packed, 3 floats per point */
int la3dm_map_insert_pointcloud(la3dm_map *m, const float *xyz, uint64_t n, const float *origin3, float ds_resolution,
                                float free_res, float max_range);
/* the same for a cloud that already lives in HBM on the map's device (n packed xyz triples); device-resident maps only */
int la3dm_map_insert_pointcloud_device(la3dm_map *m, const float *d_xyz, uint64_t n, const float *origin3, float ds_resolution,
                                       float free_resolution, float max_range);
/* insert_training_data(GPPointCloud): x,y,z,label per point */
int la3dm_map_insert_training_data(la3dm_map *m, const float *xyzy, uint64_t n);

/* split form: prepare -> (caller runs la3dm_bgk_scan_* on the packed scan) -> commit.
 * prepare returns 1 if there is work, 0 if the training set is empty. */
int la3dm_map_prepare(la3dm_map *m, const float *xyz, uint64_t n, const float *origin3, float ds_resolution,
                      float free_res, float max_range);
int la3dm_map_prepare_training_data(la3dm_map *m, const float *xyzy, uint64_t n, int ungated);
int la3dm_map_packed(la3dm_map *m, la3dm_bgk_scan *out);
int la3dm_map_commit(la3dm_map *m);
la3dm_ctx *la3dm_map_ctx(la3dm_map *m);

/* Device-resident mode (include/la3dm_hip.h, la3dm_devmap_*): the block pool lives in HBM, insert_pointcloud runs
 * start to finish on the GPU, the host blocks are a lazily refreshed mirror.  Switch while the map is empty.
 * insert_training_data and prepare/commit are refused in this mode. */
int la3dm_map_set_device_resident(la3dm_map *m, int on);
/* block-sharded insert_pointcloud over `world` replicas of the map, one per GPU (la3dm_devmap_set_shard, la3dm_hip.h);
 * the map must be device resident; world = 1 switches it off */
int la3dm_map_set_shard(la3dm_map *m, uint32_t rank, uint32_t world, la3dm_allgatherv_fn fn, void *user);
int la3dm_map_is_device_resident(const la3dm_map *m);

int la3dm_map_stats(const la3dm_map *m, la3dm_scan_stats *out);
uint64_t la3dm_map_training_size(const la3dm_map *m);
int la3dm_map_training_data(const la3dm_map *m, float *xyzy, uint64_t cap);

float la3dm_map_block_size(const la3dm_map *m);
float la3dm_map_resolution(const la3dm_map *m);
int la3dm_map_block_depth(const la3dm_map *m);
/* BGKOctoMap::set_resolution / set_block_depth (reference src/bgkoctomap/bgkoctomap.cpp:66-80, and the same pair of the
 * GP / BGK-L / BGK-LV classes): re-derive block size and voxel LUT, rebuild the device context.  Legal on an EMPTY map
 * only — the reference applies them under existing blocks and silently corrupts the map; here that is an error (-1,
 * la3dm_map_last_error).  Options set through la3dm_set_option return to their defaults. */
int la3dm_map_set_resolution(la3dm_map *m, float resolution);
int la3dm_map_set_block_depth(la3dm_map *m, int block_depth);
uint64_t la3dm_map_block_count(const la3dm_map *m);
uint64_t la3dm_map_leaf_count(const la3dm_map *m);
/* all leaves, blocks by ascending hash key, leaves in LeafIterator order */
uint64_t la3dm_map_dump_leaves(const la3dm_map *m, int64_t *block_key, int32_t *node_key, float *loc_xyz, float *size,
                               float *A, float *B, uint8_t *state, uint8_t *classified, uint64_t cap);
/* search(x, y, z): returns 1 if the block exists */
int la3dm_map_search(const la3dm_map *m, float x, float y, float z, float *A, float *B, uint8_t *state);
/* search for n points at once (packed xyz); device-resident maps answer from the device pool without a mirror refresh */
int la3dm_map_search_many(const la3dm_map *m, const float *xyz, uint64_t n, uint8_t *exists, float *A, float *B,
                          uint8_t *state);
int la3dm_map_get_bbox(const la3dm_map *m, float *lim_min3, float *lim_max3);
/* Block(center).get_index(p) / get_node / get_point (reference bgkblock.cpp:131-150) */
void la3dm_map_block_grid(const la3dm_map *m, const float *center3, const float *p3, int32_t *idx3, int32_t *node_key,
                          float *point3);
/* BGKOctoMap::RayCaster(map, start, end) driven to its end (reference bgkoctomap.h:91-214): one row per next() call —
 * voxel centre, block key, node key, valid (block exists), a copy of the node.  Returns the number of steps
 * (rows beyond `cap` are counted but not written). */
uint64_t la3dm_map_raycast(const la3dm_map *m, const float *start3, const float *end3, float *p_xyz, int64_t *block_key,
                           int32_t *node_key, uint8_t *valid, float *A, float *B, uint8_t *state, uint64_t cap);

/* Cube lists of the map: the publish loop of the static node (reference bgkoctomap_static_node.cpp:101-136) with
 * MarkerArrayPub::insert_point3d / heightMapColor (markerarray_pub.h:21-147) minus ROS.  state 1 = OCCUPIED leaves
 * coloured by height between min_z and max_z (min_z == max_z: the map's bbox), 0 = FREE leaves coloured by
 * probability; original_size 0 expands collapsed leaves into base-resolution cells (get_pruned_locs).  cells / rgba:
 * 4 floats per cell {x, y, z, size} / {r, g, b, a}; level = (int) log2(size / resolution).  Device-resident maps are
 * scanned on the GPU.  Call with NULL buffers for *count, then with buffers of that capacity. */
int la3dm_map_export_cells(const la3dm_map *m, int state, int original_size, float min_z, float max_z, float *cells,
                           float *rgba, int32_t *level, uint64_t cap, uint64_t *count);

/* host bookkeeping primitives (known-answer tests) */
int64_t la3dm_map_block_to_hash_key(const la3dm_map *m, float x, float y, float z);
void la3dm_map_hash_key_to_block(const la3dm_map *m, int64_t key, float *out3);
void la3dm_map_extended_block(const la3dm_map *m, int64_t key, int64_t *out7);
uint32_t la3dm_map_lut(const la3dm_map *m, float *xyz, uint32_t cap_entries);

#ifdef __cplusplus
}
#endif
#endif
