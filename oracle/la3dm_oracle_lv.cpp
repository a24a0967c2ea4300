// la3dm_oracle_lv.cpp — CPU restatement of la3dm's BGKLVOctoMap per-scan path (variance-aware,
// line-segment free space, per-voxel inference).
//
// *** TEST INFRASTRUCTURE ONLY *** (see la3dm_oracle.cpp).  Strict fp32 with every float->double
// promotion of the reference reproduced; build with -ffp-contract=off.
//
// Reference (paths relative to the reference checkout):
//   BGKLVOctoMap::insert_pointcloud     src/bgklvoctomap/bgklvoctomap.cpp:89-285
//   get_training_data (ray shortening)  src/bgklvoctomap/bgklvoctomap.cpp:303-423
//   beam_sample (from the end backwards) src/bgklvoctomap/bgklvoctomap.cpp:439-462
//   point_to_line_dist / covSparseLine  include/bgklvoctomap/bgklvinference.h:100-157
//   Occupancy (min_W, UNCERTAIN)        src/bgklvoctomap/bgklvoctree_node.cpp:17-77
//   OcTree (key = (depth << 28) + index) src/bgklvoctomap/bgklvoctree.cpp:9-15, 72-148
//
// PARITY PINNING: the LV node (get_prob / get_var / update / the (A, B) constructor), the 28-bit node key, the
// LUT, leaf order, prune and block hashing are pinned against the reference's own compiled LV sources
// (oracle/_ref/libla3dm_ref_lv.so = ref_harness.cpp -DLA3DM_REF_LV; golden tests/golden/ref_kat_lv.npz;
// tests/test_oracle.py::test_lv_*_against_reference_kat).  The Eigen/PCL parts (kernel arithmetic, voxel grid) and
// the R-tree gather order are restated here => **parity unpinned** for those.
//
// Row order.  The reference sums a voxel's kernel row in R-tree search order (then Eigen's GEMV
// order) — both unpinned.  Restated order: the training points are bucketed on a grid of edge
// g = 4 * resolution aligned with the blocks; a voxel visits the (2r+1)^3 buckets around its own
// (r = ceil(ell / g)) z-major, and a bucket's points in ascending index.  A ray contributes once, at
// the position of its lowest-index sample inside the voxel's box (the reference's ray_keys
// de-duplication keeps one row per ray; which sample triggers it does not change the value).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <vector>

extern int g_orc_trig_mode, g_orc_grid_sort_mode, g_orc_sum_mode;   // la3dm_oracle.cpp: the sensitivity switches (orc_set_modes)
namespace orc_eigen337 {
float psin(float x);
float pcos(float x);
}

namespace {

enum : uint8_t { LV_FREE = 0, LV_OCCUPIED = 1, LV_UNKNOWN = 2, LV_UNCERTAIN = 3, LV_PRUNED = 4 };

struct V3 {
    float x, y, z;
};
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 mulf(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
// point3f::norm(): double sqrt of a float sum; dot(): double of a float expression
inline double norm3(V3 a) { return sqrt((double)(a.x * a.x + a.y * a.y + a.z * a.z)); }
inline double dot3(V3 a, V3 b) { return (double)(a.x * b.x + a.y * b.y + a.z * b.z); }

struct Params {
    float resolution;
    int depth;
    float sf2, ell, free_thresh, occupied_thresh, var_thresh, prior_A, prior_B;
    bool original_size;
    float min_W;
    float block_size;
};

struct Node {
    uint8_t classified;
    float A, B;
    uint8_t state;
};

// src/bgklvoctomap/bgklvoctree_node.cpp:29-63 (mixed float/double kept)
float lv_prob(const Params &p, const Node &n) {
    float prob, W;
    if (n.A + n.B < p.min_W) W = p.min_W; else W = n.A + n.B;
    if (n.A > n.B)
        prob = (float)(n.A / (W - n.B) + (W - n.A - n.B) * 0.5 / (W - n.B));
    else
        prob = (float)(0.5 * (W - n.B - n.A) / (W - n.A));
    return prob;
}
float lv_var(const Params &p, const Node &n) {
    float W;
    float prob = lv_prob(p, n);
    if (n.A + n.B < p.min_W) W = p.min_W; else W = n.A + n.B;
    return (float)(n.A / W * pow(1 - prob, 2) + (W - n.A - n.B) / W * pow(0.5 - prob, 2) + n.B / W * pow(prob, 2));
}
void lv_update(const Params &p, Node &n, float ybar, float kbar) {
    n.classified = 1;
    n.A += ybar;
    n.B += kbar - ybar;
    float var = lv_var(p, n);
    if (var > p.var_thresh)
        n.state = LV_UNCERTAIN;
    else {
        float pr = lv_prob(p, n);
        n.state = pr > p.occupied_thresh ? LV_OCCUPIED : (pr < p.free_thresh ? LV_FREE : LV_UNKNOWN);
    }
}

inline int64_t block_key(const Params &p, float x, float y, float z) {
    double s = (double)p.block_size;
    return (int64_t(x / s + 524288.5) << 40) | (int64_t(y / s + 524288.5) << 20) | (int64_t(z / s + 524288.5));
}
inline V3 key_center(const Params &p, int64_t key) {
    return V3{((key >> 40) - 524288) * p.block_size, (((key >> 20) & 0xFFFFF) - 524288) * p.block_size,
              ((key & 0xFFFFF) - 524288) * p.block_size};
}

struct Block {
    V3 center;
    std::vector<std::vector<Node>> layer;
    std::vector<char> alive;
};
Block *block_new(const Params &p, V3 c) {
    Block *b = new Block;
    b->center = c;
    b->layer.resize(p.depth);
    b->alive.assign(p.depth, 1);
    size_t n = 1;
    for (int d = 0; d < p.depth; ++d, n *= 8) b->layer[d].assign(n, Node{0, p.prior_A, p.prior_B, LV_UNKNOWN});
    return b;
}
inline bool is_leaf(const Params &p, const Block &b, int d, uint32_t i) {
    if (b.alive[d] && b.layer[d][i].state != LV_PRUNED) {
        if (d + 1 < p.depth) {
            if (!b.alive[d + 1] || b.layer[d + 1][(size_t)i * 8].state == LV_PRUNED) return true;
        } else
            return true;
    }
    return false;
}
void enumerate_leaves(const Params &p, const Block &b, std::vector<uint32_t> &keys) {  // (depth << 28) + index
    keys.clear();
    std::vector<std::pair<int, uint32_t>> st;
    st.emplace_back(0, 0u);
    while (!st.empty()) {
        auto t = st.back();
        st.pop_back();
        if (is_leaf(p, b, t.first, t.second))
            keys.push_back(((uint32_t)t.first << 28) + t.second);
        else if (t.first + 1 < p.depth)
            for (uint32_t i = 0; i < 8; ++i) st.emplace_back(t.first + 1, t.second * 8 + i);
    }
}
bool block_prune(const Params &p, Block &b) {  // bgklvoctree.cpp:101-148 (same code as the BGK tree)
    bool pruned = false;
    for (int d = p.depth - 1; d > 0; --d) {
        if (!b.alive[d]) continue;
        auto &layer = b.layer[d];
        auto &parent = b.layer[d - 1];
        bool empty_layer = true;
        for (size_t g = 0; g < layer.size(); g += 8) {
            uint8_t s = layer[g].state;
            if (s == LV_UNKNOWN) { empty_layer = false; continue; }
            if (s == LV_PRUNED) continue;
            bool same = true;
            for (int i = 1; i < 8; ++i) same &= layer[g + i].state == s;
            if (same) {
                parent[g / 8].A = layer[g].A; parent[g / 8].B = layer[g].B; parent[g / 8].state = layer[g].state;
                for (int i = 0; i < 8; ++i) layer[g + i].state = LV_PRUNED;
                pruned = true;
            } else
                empty_layer = false;
        }
        if (empty_layer) { b.alive[d] = 0; std::vector<Node>().swap(b.layer[d]); }
    }
    return pruned;
}
std::vector<std::vector<V3>> build_lut(float resolution, int depth) {  // bgklvblock.cpp:7-32
    std::vector<std::vector<V3>> lut(depth);
    lut[0].push_back(V3{0, 0, 0});
    for (int d = 0; d + 1 < depth; ++d) {
        float half_size = (float)(resolution * pow(2, depth - d - 1) * 0.5f);
        for (const V3 &c : lut[d])
            for (int i = 0; i < 8; ++i)
                lut[d + 1].push_back(V3{(float)(c.x + half_size * (i & 4 ? 0.5 : -0.5)), (float)(c.y + half_size * (i & 2 ? 0.5 : -0.5)),
                                        (float)(c.z + half_size * (i & 1 ? 0.5 : -0.5))});
    }
    return lut;
}

// voxel grid as in la3dm_oracle.cpp (PCL restated) — duplicated to keep this unit self-contained
void voxel_grid(const std::vector<V3> &in, float leaf, std::vector<V3> &out) {
    out.clear();
    if (in.empty()) return;
    float inv = 1.0f / leaf;
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (const V3 &p : in) {
        if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
        mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
        mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
    }
    int64_t ex = (int64_t)((mx[0] - mn[0]) * inv) + 1, ey = (int64_t)((mx[1] - mn[1]) * inv) + 1, ez = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (ex * ey * ez > (int64_t)INT32_MAX) { out = in; return; }
    int lo[3], sp[3];
    for (int a = 0; a < 3; ++a) { lo[a] = (int)std::floor(mn[a] * inv); sp[a] = (int)std::floor(mx[a] * inv) - lo[a] + 1; }
    std::vector<std::pair<unsigned, unsigned>> iv;
    for (size_t i = 0; i < in.size(); ++i) {
        const V3 &p = in[i];
        if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
        int c0 = (int)(std::floor(p.x * inv) - (float)lo[0]), c1 = (int)(std::floor(p.y * inv) - (float)lo[1]), c2 = (int)(std::floor(p.z * inv) - (float)lo[2]);
        iv.emplace_back((unsigned)(c0 + c1 * sp[0] + c2 * sp[0] * sp[1]), (unsigned)i);
    }
    if (g_orc_grid_sort_mode) {  // pcl::VoxelGrid's unstable sort on idx alone (see la3dm_oracle.cpp)
        struct PclLess {
            bool operator()(const std::pair<unsigned, unsigned> &a, const std::pair<unsigned, unsigned> &b) const { return a.first < b.first; }
        };
        std::sort(iv.begin(), iv.end(), PclLess());
    } else {
        std::sort(iv.begin(), iv.end());
    }
    for (size_t i = 0; i < iv.size();) {
        size_t j = i;
        float sx = 0, sy = 0, sz = 0;
        for (; j < iv.size() && iv[j].first == iv[i].first; ++j) { sx += in[iv[j].second].x; sy += in[iv[j].second].y; sz += in[iv[j].second].z; }
        float n = (float)(j - i);
        out.push_back(V3{sx / n, sy / n, sz / n});
        i = j;
    }
}

struct Training {
    std::vector<V3> xy;        // sample positions (hits and ray samples)
    std::vector<int> ray_idx;  // -1 for hits
    std::vector<V3> ray0, ray1;  // free segments
    std::vector<int> ray_base;   // index in xy of each ray's first sample (= free_origin)
};

// bgklvoctomap.cpp:439-462
void beam_sample_lv(V3 hit, V3 origin, float free_resolution, std::vector<V3> &frees) {
    frees.clear();
    float x0 = origin.x, y0 = origin.y, z0 = origin.z, x = hit.x, y = hit.y, z = hit.z;
    float l = (float)sqrt((x - x0) * (x - x0) + (y - y0) * (y - y0) + (z - z0) * (z - z0));
    float nx = (x - x0) / l, ny = (y - y0) / l, nz = (z - z0) / l;
    float d = l;
    while (d > 0.0) {
        frees.push_back(V3{x0 + nx * d, y0 + ny * d, z0 + nz * d});
        d -= free_resolution;
    }
}

// bgklvoctomap.cpp:303-423
void get_training_data(const Params &P, const std::vector<V3> &cloud, V3 origin, float ds_resolution, float free_resolution,
                       float max_range, Training &T) {
    std::vector<V3> hits;
    if (ds_resolution < 0) hits = cloud; else voxel_grid(cloud, ds_resolution, hits);
    T = Training();
    int idx = 0;
    double offset = P.ell * pow(2, 0.5);
    double influence = P.ell;
    // One hit's work (the reference's loop body, :312-420) reads the whole hit list and nothing of the other hits'
    // results: in the OpenMP build the hits are processed in parallel into per-hit records, and the training set is
    // assembled from the records in hit order — the same values in the same order as the serial loop.
    struct PerHit {
        bool is_sample = false, has_ray = false;
        V3 free_origin, free_endpt;
        std::vector<V3> frees;
    };
    std::vector<PerHit> rec(hits.size());
#pragma omp parallel
    {
        std::vector<V3> nearby;
#pragma omp for schedule(dynamic, 16)
        for (int64_t hi = 0; hi < (int64_t)hits.size(); ++hi) {
            const V3 &p = hits[hi];
            PerHit &R = rec[hi];
            double l = norm3(p - origin);
            float nx = (float)((p.x - origin.x) / l), ny = (float)((p.y - origin.y) / l), nz = (float)((p.z - origin.z) / l);
            if (max_range > 0) {
                if (l < max_range) {
                    l = (float)sqrt((p.x - origin.x) * (p.x - origin.x) + (p.y - origin.y) * (p.y - origin.y) + (p.z - origin.z) * (p.z - origin.z));
                    l = l - offset;
                    R.is_sample = true;
                } else
                    l = max_range - offset;
            }
            V3 nearest_point = p;
            V3 free_endpt{(float)(origin.x + nx * l), (float)(origin.y + ny * l), (float)(origin.z + nz * l)};
            nearby.clear();
            for (const V3 &p0 : hits) {
                if (max_range > 0) {
                    double range = norm3(p0 - origin);
                    if (range > max_range) continue;
                }
                if (p.z > (offset + origin.z) && p0.z < origin.z + influence) continue;
                double dist1 = norm3(free_endpt - p0), dist2 = norm3(origin - p0);
                if (dist1 < influence) nearby.push_back(p0);
                else if (dist1 < l && dist2 < l) nearby.push_back(p0);
            }
            V3 line_vec = free_endpt - origin;
            for (const V3 &p1 : nearby) {
                double dist;
                V3 pnt_vec = p1 - origin;
                double b = dot3(pnt_vec, line_vec);
                if (b > pow(l, 2)) continue;
                V3 nearest = origin + mulf(line_vec, (float)(b / pow(norm3(line_vec), 2)));
                dist = norm3(p1 - nearest);
                if (dist < influence) {
                    nearest_point = p1;
                    l = b / norm3(line_vec);
                }
            }
            if (l < max_range / 5.0 && l / (offset - nearest_point.z) > 0) continue;
            free_endpt = V3{(float)(origin.x + nx * l), (float)(origin.y + ny * l), (float)(origin.z + nz * l)};
            V3 free_origin = origin;
            double mu = 1.0;
            if (l > influence * mu)
                free_origin = V3{(float)(origin.x + nx * influence * mu), (float)(origin.y + ny * influence * mu), (float)(origin.z + nz * influence * mu)};
            else
                free_origin = free_endpt;
            beam_sample_lv(free_endpt, free_origin, free_resolution, R.frees);
            R.has_ray = true;
            R.free_origin = free_origin;
            R.free_endpt = free_endpt;
        }
    }
    for (size_t hi = 0; hi < hits.size(); ++hi) {
        const PerHit &R = rec[hi];
        if (R.is_sample) {
            T.xy.push_back(hits[hi]);
            T.ray_idx.push_back(-1);
        }
        if (!R.has_ray) continue;
        T.ray_base.push_back((int)T.xy.size());
        T.xy.push_back(R.free_origin);
        T.ray_idx.push_back(idx);
        for (const V3 &f : R.frees) { T.xy.push_back(f); T.ray_idx.push_back(idx); }
        T.ray0.push_back(R.free_origin);
        T.ray1.push_back(R.free_endpt);
        ++idx;
    }
}


// include/bgklvoctomap/bgklvinference.h:100-134 (one point, one segment) then :143-156
inline float seg_dist(V3 p, V3 p0, V3 p1) {
    V3 line_vec = p1 - p0;
    float line_len = (float)norm3(line_vec);
    V3 pnt_vec = p - p0;
    if (line_len < 0.0001f) return (float)norm3(p - p0);
    double c1 = dot3(pnt_vec, line_vec), c2 = dot3(line_vec, line_vec);
    if (c1 <= 0) return (float)norm3(p - p0);
    if (c2 <= c1) return (float)norm3(p - p1);
    double b = c1 / c2;
    V3 nearest = p0 + mulf(line_vec, (float)b);
    return (float)norm3(p - nearest);
}
inline float cr_cosf(float t) { return g_orc_trig_mode ? orc_eigen337::pcos(t) : (float)cos((double)t); }
inline float cr_sinf(float t) { return g_orc_trig_mode ? orc_eigen337::psin(t) : (float)sin((double)t); }
inline float cov_sparse_line(float d, float ell, float sf2) {
    float r = d / ell;
    if (r > 1.0) r = 1.0f;
    float t = (r * 2.0f) * 3.1415926f;
    return (((2.0f + cr_cosf(t)) * (1.0f - r)) / 3.0f + cr_sinf(t) / (2.0f * 3.1415926f)) * sf2;  // no < 0 clamp
}

struct Stats {
    double n_hits, n_rays, n_samples, n_bbox_blocks, n_info_blocks, voxels_visited, voxel_updates, rows, t_frontend, t_infer, t_total;
};

struct Map {
    Params p;
    std::vector<std::vector<V3>> lut;
    std::unordered_map<int64_t, Block *> blocks;
    Stats st;
    ~Map() { for (auto &kv : blocks) delete kv.second; }
};

inline bool in_box(V3 lo, V3 hi, V3 q) {  // closed, rtree.h:1519-1532
    return !(lo.x > q.x || q.x > hi.x || lo.y > q.y || q.y > hi.y || lo.z > q.z || q.z > hi.z);
}

void insert_lv(Map &m, const std::vector<V3> &cloud, V3 origin, float ds_resolution, float free_res, float max_range) {
    const Params &p = m.p;
    Training T;
    if (ds_resolution > p.resolution) ds_resolution = p.resolution;  // :102-104
    get_training_data(p, cloud, origin, ds_resolution, free_res, max_range, T);
    m.st.n_hits = 0;
    for (int r : T.ray_idx) m.st.n_hits += r < 0;
    m.st.n_rays = (double)T.ray0.size();
    m.st.n_samples = (double)T.xy.size();
    if (T.xy.empty()) return;  // (the reference would dereference an empty bbox; nothing to do)
    // bbox over sample positions (x0 == x1 for every entry), :464-490
    V3 lo = T.xy[0], hi = T.xy[0];
    for (const V3 &q : T.xy) {
        lo.x = std::min(lo.x, q.x); lo.y = std::min(lo.y, q.y); lo.z = std::min(lo.z, q.z);
        hi.x = std::max(hi.x, q.x); hi.y = std::max(hi.y, q.y); hi.z = std::max(hi.z, q.z);
    }
    std::vector<int64_t> blocks;
    const float bs = p.block_size;
    for (float x = lo.x - bs; x <= hi.x + 2 * bs; x += bs)
        for (float y = lo.y - bs; y <= hi.y + 2 * bs; y += bs)
            for (float z = lo.z - bs; z <= hi.z + 2 * bs; z += bs) blocks.push_back(block_key(p, x, y, z));
    m.st.n_bbox_blocks = (double)blocks.size();

    // gather grid (see header): bucket edge g, bucket index = floor((v + block_size/2) / g) in double
    const double g = p.depth >= 3 ? 4.0 * (double)p.resolution : (double)p.block_size;
    const double half = 0.5 * (double)p.block_size;
    const int r = (int)std::ceil((double)p.ell / g);
    auto cidx = [&](float v) { return (int64_t)std::floor(((double)v + half) / g); };
    auto ckey = [](int64_t ix, int64_t iy, int64_t iz) { return ((ix + 1048576) << 42) | ((iy + 1048576) << 21) | (iz + 1048576); };
    std::unordered_map<int64_t, std::vector<int>> bucket;
    for (size_t i = 0; i < T.xy.size(); ++i) bucket[ckey(cidx(T.xy[i].x), cidx(T.xy[i].y), cidx(T.xy[i].z))].push_back((int)i);

    std::vector<int64_t> test_blocks;
    const V3 hs{p.ell, p.ell, p.ell};
    // Blocks are independent (a voxel reads the training set and writes its own node): the OpenMP build runs one task
    // per DISTINCT block key; a key the float-stepped bbox loop lists twice is visited twice by the same task, in
    // order, as the reference's serial loop does.
    std::vector<Block *> blk_of(blocks.size());
    std::vector<size_t> uniq;                 // positions of first occurrences
    std::vector<std::vector<size_t>> occ;     // per distinct key: its positions in `blocks`, ascending
    {
        std::unordered_map<int64_t, size_t> slot;
        for (size_t bi = 0; bi < blocks.size(); ++bi) {
            const int64_t key = blocks[bi];
            auto it = m.blocks.find(key);
            if (it == m.blocks.end()) it = m.blocks.emplace(key, block_new(p, key_center(p, key))).first;  // :145-146
            blk_of[bi] = it->second;
            auto sl = slot.find(key);
            if (sl == slot.end()) {
                slot.emplace(key, uniq.size());
                uniq.push_back(bi);
                occ.emplace_back(1, bi);
            } else
                occ[sl->second].push_back(bi);
        }
    }
    std::vector<char> info(blocks.size(), 0);
    double n_visited = 0, n_rows = 0, n_updates = 0;
#pragma omp parallel reduction(+ : n_visited, n_rows, n_updates)
    {
    std::vector<uint32_t> keys;
    std::vector<int> cand;
#pragma omp for schedule(dynamic, 4)
    for (int64_t ui = 0; ui < (int64_t)uniq.size(); ++ui)
    for (size_t bi : occ[ui]) {
        Block *block = blk_of[bi];
        bool has_info = false;
        enumerate_leaves(p, *block, keys);
        for (uint32_t k : keys) {
            const int d = (int)(k >> 28);
            const uint32_t idx = k & 0xFFFFFFFu;
            const float size = float(p.block_size / pow(2, d));
            if (size > p.resolution) continue;  // :157-160
            const V3 &o = m.lut[d][idx];
            const V3 c{o.x + block->center.x, o.y + block->center.y, o.z + block->center.z};
            const V3 bl = c - hs, bh = c + hs;
            n_visited += 1;
            // candidates in box, in gather order
            cand.clear();
            const int64_t cx = cidx(c.x), cy = cidx(c.y), cz = cidx(c.z);
            for (int64_t dz = -r; dz <= r; ++dz)
                for (int64_t dy = -r; dy <= r; ++dy)
                    for (int64_t dx = -r; dx <= r; ++dx) {
                        auto bt = bucket.find(ckey(cx + dx, cy + dy, cz + dz));
                        if (bt == bucket.end()) continue;
                        for (int i : bt->second)
                            if (in_box(bl, bh, T.xy[i])) cand.push_back(i);
                    }
            if (cand.empty()) continue;  // :168-175
            // one row per hit, one row per ray at its lowest-index sample in the box (:176-205)
            // sum mode 1 (orc_set_sum_mode, la3dm_oracle.cpp): the same fp32 kv and kv * y, summed in double and rounded to
            // fp32 once per voxel — the value the fp32 chains approximate, whatever the order of the rows
            const bool sum64 = g_orc_sum_mode == 1;
            double dybar = 0.0, dkbar = 0.0;
            float ybar = 0.0f, kbar = 0.0f;
            for (int i : cand) {
                const int ray = T.ray_idx[i];
                float kv, y;
                if (ray < 0) {
                    kv = cov_sparse_line(seg_dist(c, T.xy[i], T.xy[i]), p.ell, p.sf2);
                    y = 1.0f;
                } else {
                    bool first = true;
                    for (int q = T.ray_base[ray]; q < i && first; ++q)
                        if (in_box(bl, bh, T.xy[q])) first = false;
                    if (!first) continue;
                    kv = cov_sparse_line(seg_dist(c, T.ray0[ray], T.ray1[ray]), p.ell, p.sf2);
                    y = 0.0f;
                }
                if (sum64) {
                    const float ky = kv * y;
                    dybar += (double)ky;
                    dkbar += (double)kv;
                } else {
                    ybar += kv * y;
                    kbar += kv;
                }
                n_rows += 1;
            }
            if (sum64) {
                ybar = (float)dybar;
                kbar = (float)dkbar;
            }
            Node &node = block->layer[d][idx];
            if (kbar > 0.001f) {  // :236-238
                lv_update(p, node, ybar, kbar);
                n_updates += 1;
            }
            has_info = true;
        }
        info[bi] = has_info;
    }
    }
    m.st.voxels_visited += n_visited;
    m.st.rows += n_rows;
    m.st.voxel_updates += n_updates;
    for (size_t bi = 0; bi < blocks.size(); ++bi)
        if (info[bi]) test_blocks.push_back(blocks[bi]);
    m.st.n_info_blocks = (double)test_blocks.size();
    for (int64_t key : test_blocks)
        if (p.original_size) block_prune(p, *m.blocks[key]);  // :262-273
}

}  // namespace

extern "C" {

// BGKLVOctoMap(resolution, block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh, prior_A, prior_B,
//              original_size, min_W)   src/bgklvoctomap/bgklvoctomap.cpp:33-62
void *orc_lv_map_create(float resolution, int block_depth, float sf2, float ell, float free_thresh, float occupied_thresh,
                        float var_thresh, float prior_A, float prior_B, int original_size, float min_W) {
    Map *m = new Map;
    m->p = Params{resolution, block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh, prior_A, prior_B,
                  original_size != 0, min_W, (float)pow(2, block_depth - 1) * resolution};
    m->lut = build_lut(resolution, block_depth);
    std::memset(&m->st, 0, sizeof(Stats));
    return m;
}
void orc_lv_map_destroy(void *h) { delete (Map *)h; }

void orc_lv_insert_pointcloud(void *h, const float *xyz, int64_t n, const float *origin, float ds_resolution, float free_res,
                              float max_range) {
    Map *m = (Map *)h;
    std::vector<V3> cloud(n);
    for (int64_t i = 0; i < n; ++i) cloud[i] = V3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    std::memset(&m->st, 0, sizeof(Stats));
    insert_lv(*m, cloud, V3{origin[0], origin[1], origin[2]}, ds_resolution, free_res, max_range);
}
void orc_lv_stats(void *h, double *out11) { std::memcpy(out11, &((Map *)h)->st, sizeof(Stats)); }

// training data of one scan: samples (x, y, z, ray index or -1) and segments (6 floats)
int64_t orc_lv_training_data(void *h, const float *xyz, int64_t n, const float *origin, float ds_resolution, float free_res,
                             float max_range, float *xy4, int64_t cap_xy, float *rays6, int64_t cap_rays, int64_t *n_rays) {
    Map *m = (Map *)h;
    std::vector<V3> cloud(n);
    for (int64_t i = 0; i < n; ++i) cloud[i] = V3{xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    if (ds_resolution > m->p.resolution) ds_resolution = m->p.resolution;
    Training T;
    get_training_data(m->p, cloud, V3{origin[0], origin[1], origin[2]}, ds_resolution, free_res, max_range, T);
    for (int64_t i = 0; i < (int64_t)T.xy.size() && i < cap_xy; ++i) {
        xy4[4 * i] = T.xy[i].x; xy4[4 * i + 1] = T.xy[i].y; xy4[4 * i + 2] = T.xy[i].z; xy4[4 * i + 3] = (float)T.ray_idx[i];
    }
    for (int64_t i = 0; i < (int64_t)T.ray0.size() && i < cap_rays; ++i) {
        rays6[6 * i] = T.ray0[i].x; rays6[6 * i + 1] = T.ray0[i].y; rays6[6 * i + 2] = T.ray0[i].z;
        rays6[6 * i + 3] = T.ray1[i].x; rays6[6 * i + 4] = T.ray1[i].y; rays6[6 * i + 5] = T.ray1[i].z;
    }
    *n_rays = (int64_t)T.ray0.size();
    return (int64_t)T.xy.size();
}

float orc_lv_seg_dist(const float *p, const float *p0, const float *p1) {
    return seg_dist(V3{p[0], p[1], p[2]}, V3{p0[0], p0[1], p0[2]}, V3{p1[0], p1[1], p1[2]});
}
float orc_lv_kernel(float d, float ell, float sf2) { return cov_sparse_line(d, ell, sf2); }
void orc_lv_node_update(void *h, float *A, float *B, uint8_t *state, float ybar, float kbar) {
    Node n{0, *A, *B, *state};
    lv_update(((Map *)h)->p, n, ybar, kbar);
    *A = n.A; *B = n.B; *state = n.state;
}
float orc_lv_node_prob(void *h, float A, float B) { return lv_prob(((Map *)h)->p, Node{0, A, B, 0}); }
float orc_lv_node_var(void *h, float A, float B) { return lv_var(((Map *)h)->p, Node{0, A, B, 0}); }

// (A, B) constructor of the node, bgklvoctree_node.cpp:17-27: prior + (A, B), classified = false, state from var / prob
void orc_lv_node_ctor(void *h, float A, float B, float *mA, float *mB, uint8_t *state) {
    const Params &p = ((Map *)h)->p;
    Node n{0, p.prior_A + A, p.prior_B + B, LV_UNKNOWN};
    float var = lv_var(p, n);
    if (var > p.var_thresh)
        n.state = LV_UNCERTAIN;
    else {
        float pr = lv_prob(p, n);
        n.state = pr > p.occupied_thresh ? LV_OCCUPIED : (pr < p.free_thresh ? LV_FREE : LV_UNKNOWN);
    }
    *mA = n.A; *mB = n.B; *state = n.state;
}
// single-block entry points for the reference pin (tests/test_oracle.py)
int64_t orc_lv_block_to_hash_key(void *h, float x, float y, float z) { return block_key(((Map *)h)->p, x, y, z); }
void orc_lv_hash_key_to_block(void *h, int64_t key, float *out3) {
    V3 c = key_center(((Map *)h)->p, key);
    out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}
int orc_lv_lut(void *h, int depth, uint32_t index, float *out3) {
    Map *m = (Map *)h;
    if (depth < 0 || depth >= (int)m->lut.size() || index >= m->lut[depth].size()) return 0;
    const V3 &o = m->lut[depth][index];
    out3[0] = o.x; out3[1] = o.y; out3[2] = o.z;
    return 1;
}
void *orc_lv_block_new(void *h, float cx, float cy, float cz) { return block_new(((Map *)h)->p, V3{cx, cy, cz}); }
void orc_lv_block_free(void *b) { delete (Block *)b; }
int orc_lv_block_leaves(void *h, void *b, int32_t *keys, float *loc_xyz, float *sizes, int cap) {
    Map *m = (Map *)h;
    Block *blk = (Block *)b;
    std::vector<uint32_t> ks;
    enumerate_leaves(m->p, *blk, ks);
    for (int i = 0; i < (int)ks.size() && i < cap; ++i) {
        const V3 &o = m->lut[ks[i] >> 28][ks[i] & 0xFFFFFFFu];
        keys[i] = (int32_t)ks[i];
        loc_xyz[3 * i] = o.x + blk->center.x; loc_xyz[3 * i + 1] = o.y + blk->center.y; loc_xyz[3 * i + 2] = o.z + blk->center.z;
        sizes[i] = float(m->p.block_size / pow(2, ks[i] >> 28));
    }
    return (int)ks.size();
}
void orc_lv_block_update(void *h, void *b, int32_t key, float ybar, float kbar) {
    lv_update(((Map *)h)->p, ((Block *)b)->layer[(uint32_t)key >> 28][(uint32_t)key & 0xFFFFFFFu], ybar, kbar);
}
int orc_lv_block_prune(void *h, void *b) { return block_prune(((Map *)h)->p, *(Block *)b) ? 1 : 0; }
int orc_lv_block_node(void *b, int32_t key, float *A, float *B, uint8_t *state, uint8_t *classified) {
    Block *blk = (Block *)b;
    uint32_t d = (uint32_t)key >> 28, i = (uint32_t)key & 0xFFFFFFFu;
    if (d >= blk->layer.size() || !blk->alive[d]) return 0;
    const Node &n = blk->layer[d][i];
    *A = n.A; *B = n.B; *state = n.state; *classified = n.classified;
    return 1;
}

int64_t orc_lv_block_count(void *h) { return (int64_t)((Map *)h)->blocks.size(); }
// leaves of blocks that hold any classified node or any non-default state (all blocks if all != 0)
int64_t orc_lv_dump_leaves(void *h, int all, int64_t *block_key_out, int64_t *node_key, float *loc, float *size, float *A,
                           float *B, uint8_t *state, uint8_t *classified, int64_t cap) {
    Map *m = (Map *)h;
    std::vector<int64_t> bk;
    for (auto &kv : m->blocks) bk.push_back(kv.first);
    std::sort(bk.begin(), bk.end());
    int64_t n = 0;
    std::vector<uint32_t> keys;
    for (int64_t key : bk) {
        Block *b = m->blocks[key];
        enumerate_leaves(m->p, *b, keys);
        if (!all) {
            bool touched = false;
            for (uint32_t k : keys) touched |= b->layer[k >> 28][k & 0xFFFFFFFu].classified != 0 || (k >> 28) + 1 < (uint32_t)m->p.depth;
            if (!touched) continue;
        }
        for (uint32_t k : keys) {
            if (n < cap) {
                const Node &nd = b->layer[k >> 28][k & 0xFFFFFFFu];
                const V3 &o = m->lut[k >> 28][k & 0xFFFFFFFu];
                block_key_out[n] = key; node_key[n] = k;
                loc[3 * n] = o.x + b->center.x; loc[3 * n + 1] = o.y + b->center.y; loc[3 * n + 2] = o.z + b->center.z;
                size[n] = float(m->p.block_size / pow(2, k >> 28));
                A[n] = nd.A; B[n] = nd.B; state[n] = nd.state; classified[n] = nd.classified;
            }
            ++n;
        }
    }
    return n;
}

}  // extern "C"
