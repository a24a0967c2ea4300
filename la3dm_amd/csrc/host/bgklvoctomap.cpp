// bgklvoctomap.cpp — host side of the MI355X BGKLVOctoMap (variance-aware BGK with free-space
// line segments, per-voxel inference).  Front end, bucketing of the training samples, dense
// packing of the base-resolution layer and commit/prune stay here; the per-voxel box query,
// de-duplication, kernel rows and the LV node update run on the GPU (include/la3dm_hip.h,
// la3dm_bgklv_scan_*).
//
// Reference behaviour followed (file:line relative to RobustFieldAutonomyLab/la3dm):
//   constructor                       src/bgklvoctomap/bgklvoctomap.cpp:33-62
//   insert_pointcloud                 src/bgklvoctomap/bgklvoctomap.cpp:89-285
//   get_training_data                 src/bgklvoctomap/bgklvoctomap.cpp:303-423
//   beam_sample                       src/bgklvoctomap/bgklvoctomap.cpp:439-462
#include <algorithm>
#include <chrono>
#include <cstring>
#include <stdexcept>
#include <string>

#include "bgkoctomap.h"

namespace la3dm {

namespace {
double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline uint32_t layer_base(unsigned depth) { return 0x249249u & ((1u << (3u * depth)) - 1u); }

// voxel-grid centroid filter (same semantics as the BGK front end; PCL is not a dependency)
void lv_voxel_grid(const std::vector<float> &in, float leaf, std::vector<float> &out, bool pcl_order = false) {
    out.clear();
    const size_t n = in.size() / 3;
    if (n == 0) return;
    const float inv = 1.0f / leaf;
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (size_t i = 0; i < n; ++i) {
        const float *p = &in[3 * i];
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], p[a]);
            mx[a] = std::max(mx[a], p[a]);
        }
    }
    const int64_t ex = (int64_t)((mx[0] - mn[0]) * inv) + 1, ey = (int64_t)((mx[1] - mn[1]) * inv) + 1,
                  ez = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (ex * ey * ez > (int64_t)INT32_MAX) {
        out = in;
        return;
    }
    int lo[3], span[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = (int)std::floor(mn[a] * inv);
        span[a] = (int)std::floor(mx[a] * inv) - lo[a] + 1;
    }
    std::vector<uint64_t> order;
    order.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const float *p = &in[3 * i];
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        const int c0 = (int)(std::floor(p[0] * inv) - (float)lo[0]), c1 = (int)(std::floor(p[1] * inv) - (float)lo[1]),
                  c2 = (int)(std::floor(p[2] * inv) - (float)lo[2]);
        order.push_back(((uint64_t)(uint32_t)(c0 + c1 * span[0] + c2 * span[0] * span[1]) << 32) | (uint32_t)i);
    }
    if (pcl_order) {   // option "grid_order" 1: pcl::VoxelGrid's own std::sort, on the cell index alone (see host/bgkoctomap.cpp voxel_grid_filter)
        // (pairs {cell, index} as pcl::VoxelGrid's cloud_point_index_idx: same size, same comparator, hence the same introsort moves)
        std::vector<std::pair<unsigned, unsigned>> iv(order.size());
        for (size_t i = 0; i < order.size(); ++i) iv[i] = {(unsigned)(order[i] >> 32), (unsigned)order[i]};
        std::sort(iv.begin(), iv.end(), [](const std::pair<unsigned, unsigned> &a, const std::pair<unsigned, unsigned> &b) { return a.first < b.first; });
        for (size_t i = 0; i < order.size(); ++i) order[i] = ((uint64_t)iv[i].first << 32) | iv[i].second;
    } else {
        std::sort(order.begin(), order.end());
    }
    for (size_t i = 0; i < order.size();) {
        const uint32_t cell = (uint32_t)(order[i] >> 32);
        float sx = 0.f, sy = 0.f, sz = 0.f;
        size_t j = i;
        for (; j < order.size() && (uint32_t)(order[j] >> 32) == cell; ++j) {
            const float *p = &in[3 * (uint32_t)order[j]];
            sx += p[0];
            sy += p[1];
            sz += p[2];
        }
        const float c = (float)(j - i);
        out.insert(out.end(), {sx / c, sy / c, sz / c});
        i = j;
    }
}

inline int lv_code(State s) { return s == State::PRUNED ? 4 : (s == State::UNCERTAIN ? 3 : (int)s); }
inline State state_from_lv(int code) { return code == 3 ? State::UNCERTAIN : (code == 4 ? State::PRUNED : (State)code); }
}  // namespace

BGKLVOctoMap::BGKLVOctoMap(float resolution_, unsigned short block_depth_, float sf2, float ell, float free_thresh,
                           float occupied_thresh, float var_thresh, float prior_A, float prior_B, bool original_size,
                           float min_W, int device)
    : BGKOctoMap(2, resolution_, block_depth_, sf2, ell, free_thresh, occupied_thresh, var_thresh, prior_A, prior_B,
                 [&] {
                     static thread_local GPParams g;
                     g = GPParams{0.f, 0.f, 0.f, 0.f, 0.f, min_W, original_size};
                     return &g;
                 }(),
                 device) {
    if (block_depth_ > 6) throw std::runtime_error("BGKLVOctoMap: block_depth > 6 is not supported (16-bit layer index)");
    if (dmap != nullptr) la3dm_devmap_lv_set_original_size(dmap, original_size ? 1 : 0);
    std::memset(cell_min, 0, sizeof(cell_min));
    std::memset(cell_dim, 0, sizeof(cell_dim));
}

// Hits become samples with ray = -1; every kept ray contributes one free segment (shortened where it passes
// near other hits) and samples along it (first sample = segment start) that carry the ray's index.
void BGKLVOctoMap::training_data_lv(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                                    float free_resolution, float max_range) {
    bind();
    std::vector<float> packed(3 * n), hits;
    for (size_t i = 0; i < n; ++i) {
        packed[3 * i] = xyz[stride * i];
        packed[3 * i + 1] = xyz[stride * i + 1];
        packed[3 * i + 2] = xyz[stride * i + 2];
    }
    {
        int go = 0;
        const bool pcl_order = device_ctx() && la3dm_get_option(device_ctx(), "grid_order", &go) == 0 && go == 1;
        if (ds_resolution < 0) hits.swap(packed); else lv_voxel_grid(packed, ds_resolution, hits, pcl_order);
    }
    samples.clear();
    rays8.clear();
    rays6.clear();
    const size_t nh = hits.size() / 3;
    const double offset = OcTreeNode::ell * pow(2, 0.5);
    const double influence = OcTreeNode::ell;
    const float ox = origin.x(), oy = origin.y(), oz = origin.z();
    // Phase 1 (parallel over beams — each beam only reads the hit list): range gate, shortening against nearby hits,
    // the downward-ray filter and the segment end points.  Phase 2 emits samples and rays in beam order.
    struct Beam {
        uint8_t hit, skip;
        float fo[3], fe[3];
    };
    std::vector<Beam> beams(nh);
    // a small team: the loop is ~30 ms of single-core work; a team as large as the machine costs more in start-up
    // and spin-waiting (which also disturbs the HIP runtime's threads) than it saves
#pragma omp parallel num_threads(16)
    {
        std::vector<point3f> nearby;
#pragma omp for schedule(dynamic, 16)
        for (long hh = 0; hh < (long)nh; ++hh) {
            const size_t h = (size_t)hh;
            Beam &bm = beams[h];
            bm.hit = bm.skip = 0;
            const point3f p(hits[3 * h], hits[3 * h + 1], hits[3 * h + 2]);
            double l = (p - origin).norm();
            const float nx = (float)((p.x() - ox) / l), ny = (float)((p.y() - oy) / l), nz = (float)((p.z() - oz) / l);
            if (max_range > 0) {
                if (l < max_range) {
                    l = (float)sqrt((p.x() - ox) * (p.x() - ox) + (p.y() - oy) * (p.y() - oy) + (p.z() - oz) * (p.z() - oz));
                    l = l - offset;
                    bm.hit = 1;
                } else {
                    l = max_range - offset;
                }
            }
            point3f nearest_point = p;
            point3f free_endpt((float)(ox + nx * l), (float)(oy + ny * l), (float)(oz + nz * l));
            nearby.clear();
            for (size_t q = 0; q < nh; ++q) {
                const point3f p0(hits[3 * q], hits[3 * q + 1], hits[3 * q + 2]);
                if (max_range > 0 && (p0 - origin).norm() > max_range) continue;
                if (p.z() > (offset + oz) && p0.z() < oz + influence) continue;  // keeps free space above the floor
                const double dist1 = (free_endpt - p0).norm(), dist2 = (origin - p0).norm();
                if (dist1 < influence || (dist1 < l && dist2 < l)) nearby.push_back(p0);
            }
            const point3f line_vec = free_endpt - origin;
            for (const point3f &p1 : nearby) {
                const point3f pnt_vec = p1 - origin;
                const double b = (double)(pnt_vec.x() * line_vec.x() + pnt_vec.y() * line_vec.y() + pnt_vec.z() * line_vec.z());
                if (b > pow(l, 2)) continue;
                const point3f nearest = origin + line_vec * (float)(b / pow(line_vec.norm(), 2));
                if ((p1 - nearest).norm() < influence) {
                    nearest_point = p1;
                    l = b / line_vec.norm();
                }
            }
            if (l < max_range / 5.0 && l / (offset - nearest_point.z()) > 0) {  // downward rays close to the sensor
                bm.skip = 1;
                continue;
            }
            free_endpt = point3f((float)(ox + nx * l), (float)(oy + ny * l), (float)(oz + nz * l));
            point3f free_origin = free_endpt;
            if (l > influence * 1.0)
                free_origin = point3f((float)(ox + nx * influence * 1.0), (float)(oy + ny * influence * 1.0),
                                      (float)(oz + nz * influence * 1.0));
            for (int a2 = 0; a2 < 3; ++a2) {
                bm.fo[a2] = free_origin(a2);
                bm.fe[a2] = free_endpt(a2);
            }
        }
    }
    uint32_t ray = 0;
    lvst.n_hits = 0;
    for (size_t h = 0; h < nh; ++h) {
        const Beam &bm = beams[h];
        if (bm.hit) {
            samples.insert(samples.end(), {hits[3 * h], hits[3 * h + 1], hits[3 * h + 2], -1.0f});
            ++lvst.n_hits;
        }
        if (bm.skip) continue;
        const point3f free_origin(bm.fo[0], bm.fo[1], bm.fo[2]), free_endpt(bm.fe[0], bm.fe[1], bm.fe[2]);
        const uint32_t first = (uint32_t)(samples.size() / 4);
        samples.insert(samples.end(), {free_origin.x(), free_origin.y(), free_origin.z(), (float)ray});
        {  // samples from the segment end back towards its start
            const float x0 = free_origin.x(), y0 = free_origin.y(), z0 = free_origin.z();
            const float x = free_endpt.x(), y = free_endpt.y(), z = free_endpt.z();
            const float len = (float)sqrt((x - x0) * (x - x0) + (y - y0) * (y - y0) + (z - z0) * (z - z0));
            const float ux = (x - x0) / len, uy = (y - y0) / len, uz = (z - z0) / len;
            for (float d = len; d > 0.0; d -= free_resolution)
                samples.insert(samples.end(), {x0 + ux * d, y0 + uy * d, z0 + uz * d, (float)ray});
        }
        float fbits;
        std::memcpy(&fbits, &first, 4);
        rays8.insert(rays8.end(), {free_origin.x(), free_origin.y(), free_origin.z(), fbits, free_endpt.x(), free_endpt.y(),
                                   free_endpt.z(), 0.0f});
        rays6.insert(rays6.end(), {free_origin.x(), free_origin.y(), free_origin.z(), free_endpt.x(), free_endpt.y(),
                                   free_endpt.z()});
        ++ray;
    }
    lvst.n_rays = ray;
    lvst.n_samples = samples.size() / 4;
}

bool BGKLVOctoMap::prepare_lv(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                              float free_res, float max_range) {
    bind();
    ensure_host_mode();   // the split form is host-orchestrated (one download of the pool)
    device_training_stale = false;
    lvst = LVStats();
    const double t0 = wall();
    if (ds_resolution > resolution) ds_resolution = resolution;
    training_data_lv(xyz, n, stride, origin, ds_resolution, free_res, max_range);
    const double t1 = wall();
    lvst.t_frontend = t1 - t0;
    lv_blocks.clear();
    lv_center.clear();
    lv_cell0.clear();
    const size_t ns = samples.size() / 4;
    if (ns == 0) return false;

    // every block of the bounding box is allocated (float-stepped candidate loop, repeats dropped)
    float lo[3] = {samples[0], samples[1], samples[2]}, hi[3] = {samples[0], samples[1], samples[2]};
    for (size_t i = 1; i < ns; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], samples[4 * i + a]);
            hi[a] = std::max(hi[a], samples[4 * i + a]);
        }
    const float bs = get_block_size();
    std::vector<BlockHashKey> keys;
    for (float x = lo[0] - bs; x <= hi[0] + 2 * bs; x += bs)
        for (float y = lo[1] - bs; y <= hi[1] + 2 * bs; y += bs)
            for (float z = lo[2] - bs; z <= hi[2] + 2 * bs; z += bs) keys.push_back(block_to_hash_key(x, y, z));
    lvst.n_bbox_blocks = keys.size();
    std::sort(keys.begin(), keys.end());
    std::vector<uint32_t> key_mult;  // how often the candidate loop produced each distinct key
    {
        size_t w = 0;
        for (size_t i = 0; i < keys.size();) {
            size_t j = i;
            while (j < keys.size() && keys[j] == keys[i]) ++j;
            keys[w++] = keys[i];
            key_mult.push_back((uint32_t)(j - i));
            i = j;
        }
        keys.resize(w);
    }

    // bucket the samples: edge g, index floor((v + block_size/2) / g), stable counting sort
    const unsigned depth = (unsigned)get_block_depth();
    const double g = depth >= 3 ? 4.0 * (double)resolution : (double)bs;
    const double half = 0.5 * (double)bs;
    const int reach = (int)std::ceil((double)OcTreeNode::ell / g);
    auto cidx = [&](float v) { return (int64_t)std::floor(((double)v + half) / g); };
    int64_t cmin[3] = {INT64_MAX, INT64_MAX, INT64_MAX}, cmax[3] = {INT64_MIN, INT64_MIN, INT64_MIN};
    // A sample with a non-finite coordinate (a hit at the sensor itself gives a beam without direction: 0 / 0) lies in
    // no voxel's box and in no bucket; it keeps its place in `samples` (the rays refer to sample indices) but is not
    // binned.
    std::vector<int64_t> cc(3 * ns);
    std::vector<uint8_t> binned(ns, 0);
    size_t n_binned = 0;
    for (size_t i = 0; i < ns; ++i) {
        if (!std::isfinite(samples[4 * i]) || !std::isfinite(samples[4 * i + 1]) || !std::isfinite(samples[4 * i + 2])) continue;
        binned[i] = 1;
        ++n_binned;
        for (int a = 0; a < 3; ++a) {
            const int64_t c = cidx(samples[4 * i + a]);
            cc[3 * i + a] = c;
            cmin[a] = std::min(cmin[a], c);
            cmax[a] = std::max(cmax[a], c);
        }
    }
    if (n_binned == 0) return false;
    size_t ncell = 1;
    for (int a = 0; a < 3; ++a) {
        cell_min[a] = (int32_t)cmin[a];
        cell_dim[a] = (int32_t)(cmax[a] - cmin[a] + 1);
        ncell *= (size_t)cell_dim[a];
    }
    if (ncell > ((size_t)1 << 28)) throw std::runtime_error("BGKLVOctoMap: scan extent too large for the dense gather grid");
    cell_off.assign(ncell + 1, 0u);
    std::vector<uint32_t> lin(ns);
    for (size_t i = 0; i < ns; ++i) {
        if (!binned[i]) continue;
        lin[i] = (uint32_t)(((cc[3 * i + 2] - cmin[2]) * cell_dim[1] + (cc[3 * i + 1] - cmin[1])) * cell_dim[0] + (cc[3 * i] - cmin[0]));
        ++cell_off[lin[i] + 1];
    }
    for (size_t c = 0; c < ncell; ++c) cell_off[c + 1] += cell_off[c];
    sorted.resize(4 * ns);
    {
        std::vector<uint32_t> cur(cell_off.begin(), cell_off.end() - 1);
        for (size_t i = 0; i < ns; ++i) {
            if (!binned[i]) continue;
            const uint32_t d = cur[lin[i]]++;
            const uint32_t oi = (uint32_t)i;
            sorted[4 * d] = samples[4 * i];
            sorted[4 * d + 1] = samples[4 * i + 1];
            sorted[4 * d + 2] = samples[4 * i + 2];
            std::memcpy(&sorted[4 * d + 3], &oi, 4);
        }
    }
    // 3-D prefix sums of the bucket counts: "any sample within reach of this block?"
    const int bpb = depth >= 3 ? (1 << (depth - 3)) : 1;  // buckets per block edge
    auto count_in = [&](int64_t x0, int64_t x1, int64_t y0, int64_t y1, int64_t z0, int64_t z1) {
        x0 = std::max<int64_t>(x0, cmin[0]); x1 = std::min<int64_t>(x1, cmax[0]);
        y0 = std::max<int64_t>(y0, cmin[1]); y1 = std::min<int64_t>(y1, cmax[1]);
        z0 = std::max<int64_t>(z0, cmin[2]); z1 = std::min<int64_t>(z1, cmax[2]);
        if (x0 > x1 || y0 > y1 || z0 > z1) return false;
        for (int64_t z = z0; z <= z1; ++z)
            for (int64_t y = y0; y <= y1; ++y) {
                const size_t row = (size_t)(((z - cmin[2]) * cell_dim[1] + (y - cmin[1])) * cell_dim[0]);
                if (cell_off[row + (size_t)(x1 - cmin[0]) + 1] != cell_off[row + (size_t)(x0 - cmin[0])]) return true;
            }
        return false;
    };
    const size_t npb = (size_t)1 << (3 * (depth - 1));
    {   // the missing blocks' constructors (a node slab each) on the team, insertion in key order on one thread
        std::vector<BlockHashKey> missing;
        for (BlockHashKey k : keys)
            if (block_arr.find(k) == block_arr.end()) missing.push_back(k);
        std::vector<Block *> made(missing.size(), nullptr);
#pragma omp parallel for num_threads(16) schedule(static)
        for (long i = 0; i < (long)missing.size(); ++i) made[(size_t)i] = new Block(hash_key_to_block(missing[(size_t)i]));
        for (size_t i = 0; i < missing.size(); ++i) block_arr.emplace(missing[i], made[i]);
    }
    lv_all_blocks.clear();
    lv_all_center.clear();
    lv_all_cell0.clear();
    lv_mult.clear();
    lv_max_mult = 0;
    for (size_t ki = 0; ki < keys.size(); ++ki) {
        const BlockHashKey k = keys[ki];
        Block *blk = block_arr.find(k)->second;
        const point3f c = blk->get_center();
        // lowest bucket of the block: its lower corner is c - size/2, i.e. bucket floor((c - size/2 + size/2)/g)
        const int64_t b0[3] = {(int64_t)std::llround((double)c.x() / g), (int64_t)std::llround((double)c.y() / g),
                               (int64_t)std::llround((double)c.z() / g)};
        if (!count_in(b0[0] - reach, b0[0] + bpb - 1 + reach, b0[1] - reach, b0[1] + bpb - 1 + reach, b0[2] - reach,
                      b0[2] + bpb - 1 + reach))
            continue;
        if (blk->node_arr == nullptr || blk->node_arr[depth - 1] == nullptr) continue;  // finest layer fully collapsed
        lv_all_blocks.push_back(blk);
        lv_all_center.insert(lv_all_center.end(), {c.x(), c.y(), c.z()});
        lv_all_cell0.insert(lv_all_cell0.end(), {(int32_t)b0[0], (int32_t)b0[1], (int32_t)b0[2]});
        lv_mult.push_back(key_mult[ki]);
        lv_max_mult = std::max(lv_max_mult, key_mult[ki]);
    }
    lv_info.assign(lv_all_blocks.size(), 0);
    lvst.n_packed_blocks = lv_all_blocks.size();
    lvst.voxels = lv_all_blocks.size() * npb;
    lvst.voxel_updates = 0;
    const bool any = select_pass_lv(0);
    lvst.t_partition = wall() - t1;
    return any;
}

// blocks of visit number `pass` (0 = every packed block) and their node arrays as they are now
bool BGKLVOctoMap::select_pass_lv(uint32_t pass) {
    const unsigned depth = (unsigned)get_block_depth();
    const size_t npb = (size_t)1 << (3 * (depth - 1));
    lv_blocks.clear();
    lv_center.clear();
    lv_cell0.clear();
    lv_pass_index.clear();
    for (size_t b = 0; b < lv_all_blocks.size(); ++b) {
        if (lv_mult[b] <= pass) continue;
        if (lv_all_blocks[b]->node_arr == nullptr || lv_all_blocks[b]->node_arr[depth - 1] == nullptr) continue;
        lv_blocks.push_back(lv_all_blocks[b]);
        lv_center.insert(lv_center.end(), lv_all_center.begin() + 3 * b, lv_all_center.begin() + 3 * b + 3);
        lv_cell0.insert(lv_cell0.end(), lv_all_cell0.begin() + 3 * b, lv_all_cell0.begin() + 3 * b + 3);
        lv_pass_index.push_back((uint32_t)b);
    }
    lv_alpha.resize(lv_blocks.size() * npb);
    lv_beta.resize(lv_blocks.size() * npb);
    lv_state.resize(lv_blocks.size() * npb);
#pragma omp parallel for num_threads(16) schedule(static)
    for (long bb = 0; bb < (long)lv_blocks.size(); ++bb) {
        const size_t b = (size_t)bb;
        const OcTreeNode *layer = lv_blocks[b]->node_arr[depth - 1];
        for (size_t i = 0; i < npb; ++i) {
            lv_alpha[b * npb + i] = layer[i].m_A;
            lv_beta[b * npb + i] = layer[i].m_B;
            lv_state[b * npb + i] = (uint8_t)lv_code(layer[i].get_state());
        }
    }
    return !lv_blocks.empty();
}

void BGKLVOctoMap::fetch_device_training() const {
    if (!device_training_stale || dmap == nullptr) return;
    uint32_t ns = 0, nr = 0;
    la3dm_devmap_lv_training(dmap, nullptr, 0, nullptr, 0, &ns, &nr);
    samples.assign(4 * (size_t)ns, 0.0f);
    rays6.assign(6 * (size_t)nr, 0.0f);
    if (la3dm_devmap_lv_training(dmap, samples.data(), ns, rays6.data(), nr, &ns, &nr) != LA3DM_OK)
        throw std::runtime_error(std::string("BGKLVOctoMap: ") + la3dm_last_error(ctx));
    device_training_stale = false;
}

la3dm_lv_scan BGKLVOctoMap::packed_lv() {
    bind();
    la3dm_lv_scan s;
    std::memset(&s, 0, sizeof(s));
    s.samples = samples.data();
    s.sorted = sorted.data();
    s.n_samples = (uint32_t)(samples.size() / 4);
    s.rays = rays8.data();
    s.n_rays = (uint32_t)(rays8.size() / 8);
    s.cell_off = cell_off.data();
    for (int a = 0; a < 3; ++a) {
        s.cell_min[a] = cell_min[a];
        s.cell_dim[a] = cell_dim[a];
    }
    s.n_blk = (uint32_t)lv_blocks.size();
    s.blk_center = lv_center.data();
    s.blk_cell0 = lv_cell0.data();
    s.alpha = lv_alpha.data();
    s.beta = lv_beta.data();
    s.state = lv_state.data();
    return s;
}

void BGKLVOctoMap::commit_lv() {
    bind();
    const double t0 = wall();
    const unsigned depth = (unsigned)get_block_depth();
    const size_t npb = (size_t)1 << (3 * (depth - 1));
    uint64_t updates = 0;
#pragma omp parallel for num_threads(16) schedule(dynamic, 4) reduction(+ : updates)
    for (long bb = 0; bb < (long)lv_blocks.size(); ++bb) {
        const size_t b = (size_t)bb;
        OcTreeNode *layer = lv_blocks[b]->node_arr[depth - 1];
        bool info = false;
        for (size_t i = 0; i < npb; ++i) {
            const uint8_t st = lv_state[b * npb + i];
            info |= (st & LA3DM_LV_HAS_INFO) != 0;
            if (!(st & LA3DM_LEAF_UPDATED)) continue;
            layer[i].classified = true;
            layer[i].m_A = lv_alpha[b * npb + i];
            layer[i].m_B = lv_beta[b * npb + i];
            layer[i].state = state_from_lv(st & 7u);
            ++updates;
        }
        if (info) lv_info[lv_pass_index[b]] = 1;
    }
    lvst.voxel_updates += updates;
    lvst.t_commit += wall() - t0;
}

// after the last visit: prune the blocks that had information (bgklvoctomap.cpp:262-273)
void BGKLVOctoMap::finish_lv() {
    const double t0 = wall();
    uint64_t info_blocks = 0;
#pragma omp parallel for num_threads(16) schedule(dynamic, 4) reduction(+ : info_blocks)
    for (long b = 0; b < (long)lv_all_blocks.size(); ++b) {
        if (!lv_info[(size_t)b]) continue;
        ++info_blocks;
        if (OcTreeNode::original_size) lv_all_blocks[(size_t)b]->prune();
    }
    lvst.n_info_blocks = info_blocks;
    lvst.t_commit += wall() - t0;
}

void BGKLVOctoMap::insert_pointcloud(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                                     float free_res, float max_range) {
    bind();
    const double t0 = wall();
    if (ctx == nullptr) throw std::runtime_error("BGKLVOctoMap::insert_pointcloud: no device context (there is no CPU path)");
    if (dmap != nullptr) {  // device-resident mode: the whole call runs on the GPU (devmap.hip lv_insert)
        lvst = LVStats();
        if (n > 0xFFFFFFFFull) throw std::runtime_error("BGKLVOctoMap::insert_pointcloud: more than 2^32 - 1 points");
        const float o[3] = {origin.x(), origin.y(), origin.z()};
        const int rc = la3dm_devmap_insert_pointcloud_host(dmap, xyz, (uint32_t)n, (uint32_t)stride, o, ds_resolution, free_res, max_range, nullptr);
        la3dm_devmap_lv_stats ds;
        la3dm_devmap_lv_stats_get(dmap, &ds);
        lvst.n_hits = ds.n_hits; lvst.n_rays = ds.n_rays; lvst.n_samples = ds.n_samples; lvst.n_bbox_blocks = ds.n_bbox_blocks;
        lvst.n_packed_blocks = ds.n_packed_blocks; lvst.n_info_blocks = ds.n_info_blocks; lvst.voxels = ds.voxels;
        lvst.voxel_updates = ds.voxel_updates; lvst.t_frontend = ds.t_frontend; lvst.t_total = wall() - t0;
        lvst.t_device = ds.t_total - ds.t_frontend;
        mirror_dirty = true;
        device_training_stale = true;
        if (rc != LA3DM_OK) throw std::runtime_error(std::string("BGKLVOctoMap::insert_pointcloud: ") + la3dm_last_error(ctx));
        return;
    }
    if (!prepare_lv(xyz, n, stride, origin, ds_resolution, free_res, max_range)) return;
    for (uint32_t pass = 0; pass < lv_max_mult; ++pass) {
        if (pass > 0 && !select_pass_lv(pass)) continue;  // repeats of the float-stepped candidate list
        const double t1 = wall();
        la3dm_lv_scan s = packed_lv();
        if (la3dm_bgklv_scan_host(ctx, &s, nullptr) != LA3DM_OK)
            throw std::runtime_error(std::string("BGKLVOctoMap::insert_pointcloud: ") + la3dm_last_error(ctx));
        lvst.t_device += wall() - t1;
        commit_lv();
    }
    finish_lv();
    lvst.t_total = wall() - t0;
}

}  // namespace la3dm
