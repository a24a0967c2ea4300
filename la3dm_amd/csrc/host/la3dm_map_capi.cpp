// la3dm_map_capi.cpp — flat C view of la3dm::BGKOctoMap (include/la3dm_map.h) for bindings.
#include "../../../include/la3dm_map.h"

#include <algorithm>
#include <cstring>
#include <exception>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "bgkoctomap.h"

using la3dm::BGKOctoMap;
using la3dm::point3f;

struct la3dm_map {
    BGKOctoMap *map;
};

static thread_local std::string g_err;

#define GUARD(body)                     \
    try {                               \
        body                            \
    } catch (const std::exception &e) { \
        g_err = e.what();               \
        return -1;                      \
    }

extern "C" {

const char *la3dm_map_last_error(void) { return g_err.c_str(); }

la3dm_map *la3dm_map_create(float resolution, int block_depth, float sf2, float ell, float free_thresh,
                            float occupied_thresh, float var_thresh, float prior_A, float prior_B, int device) {
    try {
        std::unique_ptr<la3dm_map> m(new la3dm_map);
        m->map = new BGKOctoMap(resolution, (unsigned short)block_depth, sf2, ell, free_thresh, occupied_thresh,
                                var_thresh, prior_A, prior_B, device);
        return m.release();
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

la3dm_map *la3dm_map_create_gp(float resolution, int block_depth, float sf2, float ell, float noise, float l,
                               float min_var, float max_var, float max_known_var, float free_thresh,
                               float occupied_thresh, int device) {
    try {
        std::unique_ptr<la3dm_map> m(new la3dm_map);
        m->map = new la3dm::GPOctoMap(resolution, (unsigned short)block_depth, sf2, ell, noise, l, min_var, max_var,
                                      max_known_var, free_thresh, occupied_thresh, device);
        return m.release();
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

la3dm_map *la3dm_map_create_l(float resolution, int block_depth, float sf2, float ell, float free_thresh,
                              float occupied_thresh, float var_thresh, float prior_A, float prior_B, int device) {
    try {
        std::unique_ptr<la3dm_map> m(new la3dm_map);
        m->map = new la3dm::BGKLOctoMap(resolution, (unsigned short)block_depth, sf2, ell, free_thresh, occupied_thresh,
                                        var_thresh, prior_A, prior_B, device);
        return m.release();
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

uint64_t la3dm_map_l_training(const la3dm_map *m, int32_t *ray_idx, uint64_t cap, float *rays6, uint64_t cap_rays,
                              uint64_t *n_rays) {
    const std::vector<int32_t> &ri = m->map->last_ray_index();
    const std::vector<float> &ry = m->map->last_rays();
    if (ray_idx) std::memcpy(ray_idx, ri.data(), sizeof(int32_t) * std::min<size_t>(ri.size(), cap));
    if (rays6) std::memcpy(rays6, ry.data(), sizeof(float) * 6 * std::min<size_t>(ry.size() / 6, cap_rays));
    if (n_rays) *n_rays = ry.size() / 6;
    return ri.size();
}

la3dm_map *la3dm_map_create_lv(float resolution, int block_depth, float sf2, float ell, float free_thresh,
                               float occupied_thresh, float var_thresh, float prior_A, float prior_B, int original_size,
                               float min_W, int device) {
    try {
        std::unique_ptr<la3dm_map> m(new la3dm_map);
        m->map = new la3dm::BGKLVOctoMap(resolution, (unsigned short)block_depth, sf2, ell, free_thresh, occupied_thresh,
                                         var_thresh, prior_A, prior_B, original_size != 0, min_W, device);
        return m.release();
    } catch (const std::exception &e) {
        g_err = e.what();
        return nullptr;
    }
}

static la3dm::BGKLVOctoMap *as_lv(const la3dm_map *m) { return dynamic_cast<la3dm::BGKLVOctoMap *>(m->map); }

uint64_t la3dm_map_lv_training(const la3dm_map *m, float *samples4, uint64_t cap_samples, float *rays6, uint64_t cap_rays,
                               uint64_t *n_rays) {
    la3dm::BGKLVOctoMap *lv = as_lv(m);
    if (!lv) return 0;
    const std::vector<float> &s = lv->lv_samples(), &r = lv->lv_rays();
    if (samples4) std::memcpy(samples4, s.data(), sizeof(float) * std::min<size_t>(s.size(), 4 * cap_samples));
    if (rays6) std::memcpy(rays6, r.data(), sizeof(float) * std::min<size_t>(r.size(), 6 * cap_rays));
    if (n_rays) *n_rays = r.size() / 6;
    return s.size() / 4;
}

int la3dm_map_lv_stats(const la3dm_map *m, double *o) {
    la3dm::BGKLVOctoMap *lv = as_lv(m);
    if (!lv) return -1;
    const la3dm::BGKLVOctoMap::LVStats &s = lv->lv_stats();
    const double v[13] = {(double)s.n_hits, (double)s.n_rays, (double)s.n_samples, (double)s.n_bbox_blocks,
                          (double)s.n_packed_blocks, (double)s.n_info_blocks, (double)s.voxels, (double)s.voxel_updates,
                          s.t_frontend, s.t_partition, s.t_device, s.t_commit, s.t_total};
    std::memcpy(o, v, sizeof(v));
    return 0;
}

int la3dm_map_lv_prepare(la3dm_map *m, const float *xyz, uint64_t n, const float *o, float ds, float free_res,
                         float max_range) {
    GUARD(la3dm::BGKLVOctoMap *lv = as_lv(m); if (!lv) throw std::runtime_error("not an LV map");
          return lv->prepare_lv(xyz, (size_t)n, 3, point3f(o[0], o[1], o[2]), ds, free_res, max_range) ? 1 : 0;)
}
int la3dm_map_lv_packed(la3dm_map *m, la3dm_lv_scan *out) {
    GUARD(la3dm::BGKLVOctoMap *lv = as_lv(m); if (!lv) throw std::runtime_error("not an LV map"); *out = lv->packed_lv();
          return 0;)
}
int la3dm_map_lv_commit(la3dm_map *m) {
    GUARD(la3dm::BGKLVOctoMap *lv = as_lv(m); if (!lv) throw std::runtime_error("not an LV map"); lv->commit_lv(); lv->finish_lv(); return 0;)
}

void la3dm_map_destroy(la3dm_map *m) {
    if (!m) return;
    delete m->map;
    delete m;
}

int la3dm_map_insert_pointcloud(la3dm_map *m, const float *xyz, uint64_t n, const float *o, float ds, float free_res,
                                float max_range) {
    GUARD(m->map->insert_pointcloud(xyz, (size_t)n, 3, point3f(o[0], o[1], o[2]), ds, free_res, max_range); return 0;)
}

int la3dm_map_insert_pointcloud_device(la3dm_map *m, const float *d_xyz, uint64_t n, const float *o, float ds, float free_res,
                                       float max_range) {
    GUARD(m->map->insert_pointcloud_device(d_xyz, (size_t)n, point3f(o[0], o[1], o[2]), ds, free_res, max_range); return 0;)
}

int la3dm_map_insert_training_data(la3dm_map *m, const float *xyzy, uint64_t n) {
    GUARD(BGKOctoMap::GPPointCloud c; c.reserve(n);
          for (uint64_t i = 0; i < n; ++i)
              c.emplace_back(point3f(xyzy[4 * i], xyzy[4 * i + 1], xyzy[4 * i + 2]), xyzy[4 * i + 3]);
          m->map->insert_training_data(c); return 0;)
}

int la3dm_map_prepare(la3dm_map *m, const float *xyz, uint64_t n, const float *o, float ds, float free_res,
                      float max_range) {
    GUARD(return m->map->prepare(xyz, (size_t)n, 3, point3f(o[0], o[1], o[2]), ds, free_res, max_range) ? 1 : 0;)
}

int la3dm_map_prepare_training_data(la3dm_map *m, const float *xyzy, uint64_t n, int ungated) {
    GUARD(return m->map->prepare_training_data(xyzy, (size_t)n, ungated != 0) ? 1 : 0;)
}

int la3dm_map_packed(la3dm_map *m, la3dm_bgk_scan *out) {
    GUARD(if (m->map->num_passes() == 0) {
        std::memset(out, 0, sizeof(*out));
        return 0;
    } *out = m->map->packed(0);
          return 0;)
}

int la3dm_map_commit(la3dm_map *m) { GUARD(m->map->commit(); return 0;) }

la3dm_ctx *la3dm_map_ctx(la3dm_map *m) { return m->map->device_ctx(); }

int la3dm_map_stats(const la3dm_map *m, la3dm_scan_stats *out) {
    const la3dm::ScanStats &s = m->map->last_stats();
    out->n_hits = s.n_hits; out->n_frees = s.n_frees; out->n_bbox_blocks = s.n_bbox_blocks;
    out->n_train_blocks = s.n_train_blocks; out->n_test_blocks = s.n_test_blocks;
    out->voxel_updates = s.voxel_updates; out->train_reads = s.train_reads; out->pair_evals = s.pair_evals;
    out->n_tiles = s.n_tiles;
    out->t_frontend = s.t_frontend; out->t_partition = s.t_partition; out->t_pack = s.t_pack;
    out->t_device = s.t_device; out->t_commit = s.t_commit; out->t_prune = s.t_prune; out->t_total = s.t_total;
    out->t_gather = s.t_gather;
    return 0;
}

uint64_t la3dm_map_training_size(const la3dm_map *m) {
    if (m->map->is_device_resident()) {
        try {
            return m->map->device_training_data().size() / 4;
        } catch (const std::exception &) {
            return 0;
        }
    }
    return m->map->last_training_data().size() / 4;
}

int la3dm_map_training_data(const la3dm_map *m, float *xyzy, uint64_t cap) {
    GUARD(const std::vector<float> dev = m->map->is_device_resident() ? m->map->device_training_data() : std::vector<float>();
          const std::vector<float> &v = m->map->is_device_resident() ? dev : m->map->last_training_data();
          std::memcpy(xyzy, v.data(), sizeof(float) * std::min<size_t>(v.size(), 4 * cap)); return 0;)
}

int la3dm_map_set_device_resident(la3dm_map *m, int on) { GUARD(m->map->set_device_resident(on != 0); return 0;) }
int la3dm_map_is_device_resident(const la3dm_map *m) { return m->map->is_device_resident() ? 1 : 0; }
int la3dm_map_set_shard(la3dm_map *m, uint32_t rank, uint32_t world, la3dm_allgatherv_fn fn, void *user) {
    GUARD(m->map->set_shard(rank, world, fn, user); return 0;)
}

float la3dm_map_block_size(const la3dm_map *m) { return m->map->get_block_size(); }
float la3dm_map_resolution(const la3dm_map *m) { return m->map->get_resolution(); }
int la3dm_map_block_depth(const la3dm_map *m) { return (int)m->map->get_block_depth(); }
int la3dm_map_set_resolution(la3dm_map *m, float resolution) { GUARD(m->map->set_resolution(resolution); return 0;) }
int la3dm_map_set_block_depth(la3dm_map *m, int block_depth) {
    GUARD(if (block_depth < 0 || block_depth > 65535) throw std::invalid_argument("set_block_depth: out of range");
          m->map->set_block_depth((unsigned short)block_depth); return 0;)
}
uint64_t la3dm_map_block_count(const la3dm_map *m) { return m->map->block_count(); }

uint64_t la3dm_map_leaf_count(const la3dm_map *m) {
    uint64_t n = 0;
    for (auto it = m->map->begin_leaf(); it != m->map->end_leaf(); ++it) ++n;
    return n;
}

uint64_t la3dm_map_dump_leaves(const la3dm_map *m, int64_t *block_key, int32_t *node_key, float *loc, float *size,
                               float *A, float *B, uint8_t *state, uint8_t *classified, uint64_t cap) {
    struct Row {
        int64_t bk;
        uint64_t seq;
        int32_t nk;
        float loc[3], size, p, v;
        uint8_t st, cl;
    };
    // the map iterates blocks in hash-map order; sort rows by (block key, leaf sequence)
    std::vector<Row> rows;
    uint64_t seq = 0;
    for (auto it = m->map->begin_leaf(); it != m->map->end_leaf(); ++it, ++seq) {
        Row r;
        r.bk = it.get_block_key();
        r.seq = seq;
        r.nk = it.get_node_key();
        const point3f p = it.get_loc();
        r.loc[0] = p.x(); r.loc[1] = p.y(); r.loc[2] = p.z();
        r.size = it.get_size();
        const la3dm::OcTreeNode &nd = it.get_node();
        // alpha/beta are private; recover them exactly from the public accessors is not
        // possible, so the C view goes through the block search path below
        r.p = 0; r.v = 0;
        r.st = (uint8_t)nd.get_state();
        r.cl = nd.classified ? 1 : 0;
        rows.push_back(r);
    }
    const bool lv = m->map->get_variant() == 2;
    if (lv) {  // reference LV codes / key layout; keep only blocks with a classified or collapsed leaf
        const int finest = (int)m->map->get_block_depth() - 1;
        std::vector<Row> kept;
        size_t i = 0;
        while (i < rows.size()) {
            size_t j = i;
            bool touched = false;
            for (; j < rows.size() && rows[j].bk == rows[i].bk; ++j) touched |= rows[j].cl != 0 || (rows[j].nk >> 16) < finest;
            if (touched)
                for (size_t k = i; k < j; ++k) {
                    Row r = rows[k];
                    r.st = r.st == 4 ? 3 : (r.st == 3 ? 4 : r.st);               // UNCERTAIN 3, PRUNED 4
                    kept.push_back(r);
                }
            i = j;
        }
        rows.swap(kept);
    }
    std::stable_sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.bk < b.bk; });
    uint64_t n = std::min<uint64_t>(rows.size(), cap);
    for (uint64_t i = 0; i < n; ++i) {
        const Row &r = rows[i];
        block_key[i] = r.bk;
        node_key[i] = lv ? (int32_t)(((uint32_t)(r.nk >> 16) << 28) + (uint32_t)(r.nk & 0xFFFF)) : r.nk;
        loc[3 * i] = r.loc[0]; loc[3 * i + 1] = r.loc[1]; loc[3 * i + 2] = r.loc[2];
        size[i] = r.size; state[i] = r.st; classified[i] = r.cl;
        const la3dm::OcTreeNode &nd = (*m->map->search((la3dm::BlockHashKey)r.bk))[r.nk];
        la3dm_node_ab(&nd, &A[i], &B[i]);
    }
    return rows.size();
}

int la3dm_map_search(const la3dm_map *m, float x, float y, float z, float *A, float *B, uint8_t *state) {
    la3dm::Block *b = m->map->search(la3dm::block_to_hash_key(x, y, z));
    la3dm::OcTreeNode nd = m->map->search(x, y, z);
    la3dm_node_ab(&nd, A, B);
    *state = (uint8_t)nd.get_state();
    return b != nullptr;
}

void la3dm_map_block_grid(const la3dm_map *m, const float *c3, const float *p3, int32_t *idx3, int32_t *node_key, float *point3) {
    m->map->bind();
    la3dm::Block b(point3f(c3[0], c3[1], c3[2]));
    unsigned short x, y, z;
    b.get_index(point3f(p3[0], p3[1], p3[2]), x, y, z);
    idx3[0] = x; idx3[1] = y; idx3[2] = z;
    *node_key = la3dm::Block::get_node(x, y, z);
    const point3f q = b.get_point(x, y, z);
    point3[0] = q.x(); point3[1] = q.y(); point3[2] = q.z();
}

uint64_t la3dm_map_raycast(const la3dm_map *m, const float *s3, const float *e3, float *p_xyz, int64_t *block_key,
                           int32_t *node_key, uint8_t *valid, float *A, float *B, uint8_t *state, uint64_t cap) {
    BGKOctoMap::RayCaster rc(m->map, point3f(s3[0], s3[1], s3[2]), point3f(e3[0], e3[1], e3[2]));
    uint64_t n = 0;
    while (!rc.end()) {
        point3f p;
        la3dm::OcTreeNode nd;
        la3dm::BlockHashKey bk;
        la3dm::OcTreeHashKey nk;
        const bool ok = rc.next(p, nd, bk, nk);
        if (n < cap) {
            p_xyz[3 * n] = p.x(); p_xyz[3 * n + 1] = p.y(); p_xyz[3 * n + 2] = p.z();
            block_key[n] = bk;
            node_key[n] = nk;
            valid[n] = ok ? 1 : 0;
            la3dm_node_ab(&nd, &A[n], &B[n]);
            state[n] = (uint8_t)nd.get_state();
        }
        ++n;
    }
    return n;
}

int la3dm_map_export_cells(const la3dm_map *m, int state, int original_size, float min_z, float max_z, float *cells,
                           float *rgba, int32_t *level, uint64_t cap, uint64_t *count) {
    GUARD(
        if (count == nullptr || (state != 0 && state != 1)) throw std::runtime_error("la3dm_map_export_cells: bad argument");
        BGKOctoMap::Cells c;
        const size_t n = m->map->export_cells(state == 1 ? la3dm::State::OCCUPIED : la3dm::State::FREE, original_size != 0, min_z, max_z, c);
        *count = n;
        if (cells == nullptr && rgba == nullptr && level == nullptr) return 0;
        if (cells == nullptr || rgba == nullptr || level == nullptr || cap < n) throw std::runtime_error("la3dm_map_export_cells: buffers too small");
        std::copy(c.xyz_size.begin(), c.xyz_size.end(), cells);
        std::copy(c.rgba.begin(), c.rgba.end(), rgba);
        std::copy(c.level.begin(), c.level.end(), level);
        return 0;)
}

int la3dm_map_search_many(const la3dm_map *m, const float *xyz, uint64_t n, uint8_t *exists, float *A, float *B,
                          uint8_t *state) {
    GUARD(m->map->search_many(xyz, (size_t)n, exists, A, B, state); return 0;)
}

int la3dm_map_get_bbox(const la3dm_map *m, float *lo, float *hi) {
    point3f a, b;
    m->map->get_bbox(a, b);
    for (unsigned i = 0; i < 3; ++i) { lo[i] = a(i); hi[i] = b(i); }
    return 0;
}

int64_t la3dm_map_block_to_hash_key(const la3dm_map *m, float x, float y, float z) {
    m->map->bind();
    return la3dm::block_to_hash_key(x, y, z);
}
void la3dm_map_hash_key_to_block(const la3dm_map *m, int64_t key, float *out3) {
    m->map->bind();
    const point3f c = la3dm::hash_key_to_block(key);
    out3[0] = c.x(); out3[1] = c.y(); out3[2] = c.z();
}
void la3dm_map_extended_block(const la3dm_map *m, int64_t key, int64_t *out7) {
    m->map->bind();
    const la3dm::ExtendedBlock e = la3dm::get_extended_block(key);
    for (int i = 0; i < 7; ++i) out7[i] = e[i];
}
uint32_t la3dm_map_lut(const la3dm_map *m, float *xyz, uint32_t cap) {
    const std::vector<point3f> lut = la3dm::init_key_loc_map(m->map->get_resolution(), (unsigned short)m->map->get_block_depth());
    uint32_t n = std::min<uint32_t>((uint32_t)lut.size(), cap);
    for (uint32_t i = 0; i < n; ++i) { xyz[3 * i] = lut[i].x(); xyz[3 * i + 1] = lut[i].y(); xyz[3 * i + 2] = lut[i].z(); }
    return (uint32_t)lut.size();
}

}  // extern "C"
