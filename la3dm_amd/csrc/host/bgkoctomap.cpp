// bgkoctomap.cpp — host side of the MI355X BGKOctoMap: node/octree/block bookkeeping,
// training-set front end, block partition (counting sort + closed-box rule), scan packing,
// commit + prune.  Kernel evaluation and fusion happen on the GPU via include/la3dm_hip.h.
//
// Reference behaviour followed (file:line relative to RobustFieldAutonomyLab/la3dm):
//   constructor / statics              src/bgkoctomap/bgkoctomap.cpp:31-56
//   insert_pointcloud stages A..G      src/bgkoctomap/bgkoctomap.cpp:214-366
//   get_training_data / beam_sample    src/bgkoctomap/bgkoctomap.cpp:383-458
//   bbox / get_blocks_in_bbox          src/bgkoctomap/bgkoctomap.cpp:464-495
//   closed-box gather                  src/bgkoctomap/bgkoctomap.cpp:497-552, include/common/rtree.h:1519-1532
//   hashing / LUT / extended block     src/bgkoctomap/bgkblock.cpp:7-32, 73-130
//   OcTree leaves / prune              src/bgkoctomap/bgkoctree.cpp:72-148, include/bgkoctomap/bgkoctree.h:62-147
//   Occupancy                          src/bgkoctomap/bgkoctree_node.cpp:15-44
#include "bgkoctomap.h"

#include <algorithm>
#include <parallel/algorithm>
#include <omp.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <stdexcept>
#include <string>

namespace la3dm {

namespace {
constexpr int kHostThreads = 16;  // OpenMP team of the host-side loops (start-up and spin-wait cost grow with the team)
double wall() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
inline uint32_t layer_base(unsigned depth) { return 0x249249u & ((1u << (3u * depth)) - 1u); }
}  // namespace

// ------------------------------------------------------------------ Occupancy
float Occupancy::sf2 = 1.0f;
float Occupancy::ell = 1.0f;
float Occupancy::free_thresh = 0.3f;
float Occupancy::occupied_thresh = 0.7f;
float Occupancy::var_thresh = 1000.0f;
float Occupancy::prior_A = 0.5f;
float Occupancy::prior_B = 0.5f;
int Occupancy::variant = 0;
float Occupancy::init_A = 0.5f;
float Occupancy::init_B = 0.5f;
float Occupancy::noise = 0.01f;
float Occupancy::l = 100.f;
float Occupancy::max_ivar = 1000.0f;
float Occupancy::min_ivar = 0.001f;
float Occupancy::min_known_ivar = 10.0f;

float Occupancy::min_W = 0.1f;
bool Occupancy::original_size = true;

float Occupancy::get_var() const {
    if (variant == 1) return 1.0f / m_B;
    if (variant == 2) {  // bgklvoctree_node.cpp:49-63
        const float prob = get_prob();
        const float W = (m_A + m_B < min_W) ? min_W : m_A + m_B;
        return (float)(m_A / W * pow(1 - prob, 2) + (W - m_A - m_B) / W * pow(0.5 - prob, 2) + m_B / W * pow(prob, 2));
    }
    return (m_A * m_B) / ((m_A + m_B) * (m_A + m_B) * (m_A + m_B + 1.0f));
}

float Occupancy::get_prob() const {
    if (variant == 2) {  // bgklvoctree_node.cpp:29-47
        const float W = (m_A + m_B < min_W) ? min_W : m_A + m_B;
        if (m_A > m_B) return (float)(m_A / (W - m_B) + (W - m_A - m_B) * 0.5 / (W - m_B));
        return (float)(0.5 * (W - m_B - m_A) / (W - m_A));
    }
    if (variant == 1) return 1.0f / (1.0f + (float)exp((double)(-l * m_A / max_ivar)));  // gpoctree_node.cpp:31-34
    return m_A / (m_A + m_B);
}

void Occupancy::classify() {
    if (variant == 1) {  // gpoctree_node.cpp:40-48
        if (m_B < min_known_ivar) {
            state = State::UNKNOWN;
            return;
        }
        m_B = m_B > max_ivar ? max_ivar : m_B;
        const float p = get_prob();
        state = p > occupied_thresh ? State::OCCUPIED : (p < free_thresh ? State::FREE : State::UNKNOWN);
        return;
    }
    if (get_var() > var_thresh) {
        state = variant == 2 ? State::UNCERTAIN : State::UNKNOWN;
        return;
    }
    float p = get_prob();
    state = p > occupied_thresh ? State::OCCUPIED : (p < free_thresh ? State::FREE : State::UNKNOWN);
}

Occupancy::Occupancy(float A, float B) : classified(false) {
    if (variant == 1) {  // Occupancy(m, var), gpoctree_node.cpp:19-29
        m_A = A / B;
        m_B = 1.0f / B;
    } else {
        m_A = prior_A + A;
        m_B = prior_B + B;
    }
    classify();
}

std::ofstream &operator<<(std::ofstream &os, const Occupancy &oc) {
    os.write((const char *)&oc.m_A, sizeof(oc.m_A));
    os.write((const char *)&oc.m_B, sizeof(oc.m_B));
    return os;
}
std::ifstream &operator>>(std::ifstream &is, Occupancy &oc) {
    float A, B;
    is.read((char *)&A, sizeof(A));
    is.read((char *)&B, sizeof(B));
    oc = OcTreeNode(A, B);
    return is;
}
std::ostream &operator<<(std::ostream &os, const Occupancy &oc) { return os << '(' << oc.m_A << ' ' << oc.m_B << ' ' << oc.get_prob() << ')'; }

void Occupancy::update(float ybar, float kbar) {
    classified = true;
    if (variant == 1) {  // BCM: update(new_m, new_var), gpoctree_node.cpp:36-39
        m_B = (float)((double)m_B + (1.0 / (double)kbar - (double)sf2));
        m_A += ybar / kbar;
    } else {
        m_A += ybar;
        m_B += kbar - ybar;
    }
    classify();
}

// --------------------------------------------------------------------- OcTree
unsigned short OcTree::max_depth = 0;

OcTree::OcTree() : node_arr(nullptr), slab(nullptr), ever_pruned(false) {
    if (max_depth == 0) return;
    size_t total = layer_base(max_depth);
    slab = new OcTreeNode[total]();
    node_arr = new OcTreeNode *[max_depth];
    for (unsigned d = 0; d < max_depth; ++d) node_arr[d] = slab + layer_base(d);
}

OcTree::~OcTree() {
    delete[] node_arr;
    delete[] slab;
}

bool OcTree::is_leaf(unsigned short depth, unsigned short index) const {
    if (node_arr == nullptr || node_arr[depth] == nullptr) return false;
    if (node_arr[depth][index].get_state() == State::PRUNED) return false;
    if (depth + 1 >= max_depth) return true;
    return node_arr[depth + 1] == nullptr || node_arr[depth + 1][index * 8].get_state() == State::PRUNED;
}

void OcTree::load_nodes(const float *A, const float *B, const uint8_t *S, size_t n) {
    if (slab == nullptr || n != layer_base(max_depth)) throw std::runtime_error("OcTree::load_nodes: node count mismatch");
    bool pruned = false;
    for (size_t i = 0; i < n; ++i) {
        OcTreeNode &nd = slab[i];
        nd.m_A = A[i];
        nd.m_B = B[i];
        nd.state = (State)(S[i] & 7u);
        nd.classified = (S[i] & 0x80u) != 0;
        pruned |= nd.state == State::PRUNED;
    }
    for (unsigned d = 0; d < max_depth; ++d) node_arr[d] = slab + layer_base(d);  // layers are never retired here
    ever_pruned = pruned;
}

bool OcTree::is_leaf(OcTreeHashKey key) const { return is_leaf((unsigned short)(key >> 16), (unsigned short)(key & 0xFFFF)); }

bool OcTree::search(OcTreeHashKey key) const {
    unsigned d = key >> 16, i = key & 0xFFFF;
    return node_arr != nullptr && node_arr[d] != nullptr && node_arr[d][i].get_state() != State::PRUNED;
}

// Nodes stay addressable after their layer was retired by prune() (the reference frees the layer and would read
// freed memory here): the slab keeps them, with state PRUNED.
OcTreeNode &OcTree::operator[](OcTreeHashKey key) const { return slab[layer_base((unsigned)(key >> 16)) + (key & 0xFFFF)]; }

// Bottom-up collapse of sibling groups that share one non-UNKNOWN state; the parent becomes a
// copy of child 0 (not an average) and a layer with nothing left to collapse is retired.
bool OcTree::prune() {
    if (node_arr == nullptr) return false;
    bool any = false;
    for (int depth = max_depth - 1; depth > 0; --depth) {
        OcTreeNode *layer = node_arr[depth];
        if (layer == nullptr) continue;
        OcTreeNode *parents = node_arr[depth - 1];
        const unsigned n = 1u << (3 * depth);
        bool retire = true;
        for (unsigned g = 0; g < n; g += 8) {
            const State s0 = layer[g].get_state();
            if (s0 == State::PRUNED) continue;
            if (s0 == State::UNKNOWN) {
                retire = false;
                continue;
            }
            bool same = true;
            for (unsigned c = 1; c < 8; ++c) same &= (layer[g + c].get_state() == s0);
            if (!same) {
                retire = false;
                continue;
            }
            parents[g >> 3] = layer[g];
            for (unsigned c = 0; c < 8; ++c) layer[g + c].prune();
            any = true;
        }
        if (retire) node_arr[depth] = nullptr;
    }
    if (any) ever_pruned = true;
    return any;
}

OcTree::LeafIterator::LeafIterator(const OcTree *t) : tree(t != nullptr && t->node_arr != nullptr ? t : nullptr), top(0) {
    if (tree == nullptr) return;
    stack[top++] = node_to_hash_key(0, 0);
    settle();
    if (top == 0) tree = nullptr;
}

void OcTree::LeafIterator::settle() {
    while (top > 0 && !tree->is_leaf(stack[top - 1])) {
        const OcTreeHashKey k = stack[--top];
        const unsigned d = k >> 16, i = k & 0xFFFF;
        if (d + 1 >= OcTree::max_depth) continue;  // pruned node at the last layer
        for (unsigned c = 0; c < 8; ++c) stack[top++] = node_to_hash_key(d + 1, i * 8 + c);
    }
}

OcTree::LeafIterator &OcTree::LeafIterator::operator++() {
    if (top > 0) {
        --top;
        settle();
    }
    if (top == 0) tree = nullptr;
    return *this;
}

void OcTree::collect_leaves(std::vector<uint32_t> &keys) const {
    if (node_arr == nullptr) return;
    if (!ever_pruned) {  // untouched tree: the whole finest layer, highest index first
        const unsigned d = max_depth - 1, n = 1u << (3 * d);
        for (unsigned i = n; i-- > 0;) keys.push_back((d << 16) + i);
        return;
    }
    for (LeafIterator it(this); it != LeafIterator(); ++it) keys.push_back((uint32_t)it.get_hash_key());
}

// ---------------------------------------------------------------------- Block
float Block::resolution = 0.1f;
float Block::size = 0.8f;
std::vector<point3f> Block::key_loc_map;

std::vector<point3f> init_key_loc_map(float resolution, unsigned short max_depth) {
    // Layer by layer; the child offsets are formed in double and rounded once, exactly as
    // the reference's breadth-first construction does (bgkblock.cpp:14-28).
    std::vector<point3f> lut(layer_base(max_depth));
    if (max_depth == 0) return lut;
    lut[0] = point3f(0.0f, 0.0f, 0.0f);
    for (unsigned d = 0; d + 1 < max_depth; ++d) {
        const float half_size = (float)(resolution * pow(2, max_depth - d - 1) * 0.5f);
        const unsigned n = 1u << (3 * d);
        for (unsigned i = 0; i < n; ++i) {
            const point3f c = lut[layer_base(d) + i];
            for (unsigned k = 0; k < 8; ++k)
                lut[layer_base(d + 1) + 8 * i + k] =
                    point3f((float)(c.x() + half_size * (k & 4 ? 0.5 : -0.5)), (float)(c.y() + half_size * (k & 2 ? 0.5 : -0.5)),
                            (float)(c.z() + half_size * (k & 1 ? 0.5 : -0.5)));
        }
    }
    return lut;
}

BlockHashKey block_to_hash_key(point3f c) { return block_to_hash_key(c.x(), c.y(), c.z()); }

BlockHashKey block_to_hash_key(float x, float y, float z) {
    const double s = (double)Block::size;
    return (int64_t(x / s + 524288.5) << 40) | (int64_t(y / s + 524288.5) << 20) | int64_t(z / s + 524288.5);
}

point3f hash_key_to_block(BlockHashKey key) {
    return point3f(((key >> 40) - 524288) * Block::size, (((key >> 20) & 0xFFFFF) - 524288) * Block::size,
                   ((key & 0xFFFFF) - 524288) * Block::size);
}

static ExtendedBlock extended_from(float x, float y, float z, BlockHashKey self, float s) {
    ExtendedBlock e;
    e[0] = self;
    e[1] = block_to_hash_key(s + x, 0 + y, 0 + z);
    e[2] = block_to_hash_key(-s + x, 0 + y, 0 + z);
    e[3] = block_to_hash_key(0 + x, s + y, 0 + z);
    e[4] = block_to_hash_key(0 + x, -s + y, 0 + z);
    e[5] = block_to_hash_key(0 + x, 0 + y, s + z);
    e[6] = block_to_hash_key(0 + x, 0 + y, -s + z);
    return e;
}

ExtendedBlock get_extended_block(BlockHashKey key) {
    const point3f c = hash_key_to_block(key);
    return extended_from(c.x(), c.y(), c.z(), key, Block::size);
}

ExtendedBlock Block::get_extended_block() const {
    return extended_from(center.x(), center.y(), center.z(), block_to_hash_key(center.x(), center.y(), center.z()), size);
}

OcTreeNode &Block::search(point3f p) const {
    // finest-layer voxel containing p; child bit 4 -> +x, 2 -> +y, 1 -> +z at every level
    const int cells = 1 << (max_depth - 1);
    auto cell = [&](float v, float c) {
        int i = (int)std::floor((v - c) / resolution + cells / 2.0f);
        return std::max(0, std::min(i, cells - 1));
    };
    const int ix = cell(p.x(), center.x()), iy = cell(p.y(), center.y()), iz = cell(p.z(), center.z());
    unsigned index = 0;
    for (int level = max_depth - 2; level >= 0; --level)
        index = index * 8 + ((((ix >> level) & 1) << 2) | (((iy >> level) & 1) << 1) | ((iz >> level) & 1));
    return slab[layer_base(max_depth - 1) + index];  // also valid when prune() retired the finest layer
}

void Block::get_index(const point3f &p, unsigned short &x, unsigned short &y, unsigned short &z) const {
    const int cells = cell_num();
    auto cell = [&](float v, float c) {
        const int i = (int)((v - c) / resolution + cells / 2);
        return (unsigned short)std::max(0, std::min(i, cells - 1));
    };
    x = cell(p.x(), center.x());
    y = cell(p.y(), center.y());
    z = cell(p.z(), center.z());
}

OcTreeHashKey Block::get_node(unsigned short x, unsigned short y, unsigned short z) {
    unsigned index = 0;
    for (int level = max_depth - 2; level >= 0; --level)
        index = index * 8 + ((((x >> level) & 1u) << 2) | (((y >> level) & 1u) << 1) | ((z >> level) & 1u));
    return node_to_hash_key((unsigned short)(max_depth - 1), (unsigned short)index);
}

// ------------------------------------------------------------------ RayCaster
BGKOctoMap::RayCaster::RayCaster(const BGKOctoMap *m, const point3f &start, const point3f &end_)
    : map(m), block(nullptr), n(0), lim(1 << (m->block_depth - 1)) {
    m->bind();
    key = block_to_hash_key(start);
    block = map->search(key);
    if (block == nullptr) return;  // the walk only starts inside an existing block
    unsigned short x, y, z;
    block->get_index(start, x, y, z);
    idx[0] = x;
    idx[1] = y;
    idx[2] = z;
    block_center = block->get_center();
    current_p = start;
    const float res = map->resolution;
    const int a0[3] = {(int)(start.x() / res), (int)(start.y() / res), (int)(start.z() / res)};
    const int a1[3] = {(int)(end_.x() / res), (int)(end_.y() / res), (int)(end_.z() / res)};
    int d[3];
    for (int a = 0; a < 3; ++a) {
        d[a] = std::abs(a1[a] - a0[a]);
        inc[a] = a1[a] > a0[a] ? 1 : (a1[a] == a0[a] ? 0 : -1);
    }
    n = 1 + d[0] + d[1] + d[2];
    err_xy = d[0] - d[1];
    err_xz = d[0] - d[2];
    err_yz = d[1] - d[2];
    for (int a = 0; a < 3; ++a) d2[a] = 2 * d[a];
}

// leave the block through the face of `axis`: neighbour centre = centre +- size on that axis, re-hashed
void BGKOctoMap::RayCaster::enter_block(int axis, int step) {
    block_center(axis) += step * map->block_size;
    key = block_to_hash_key(block_center);
    block = map->search(key);
    idx[axis] = step > 0 ? 0 : lim - 1;
}

bool BGKOctoMap::RayCaster::next(point3f &p, OcTreeNode &node, BlockHashKey &block_key, OcTreeHashKey &node_key) {
    node_key = Block::get_node((unsigned short)idx[0], (unsigned short)idx[1], (unsigned short)idx[2]);
    block_key = key;
    const bool valid = block != nullptr;
    if (valid) {
        node = (*block)[node_key];
        current_p = block->get_point((unsigned short)idx[0], (unsigned short)idx[1], (unsigned short)idx[2]);
    }
    p = current_p;
    const float res = map->resolution;
    auto step = [&](int axis) {
        idx[axis] += inc[axis];
        current_p(axis) += inc[axis] * res;
        if (idx[axis] >= lim || idx[axis] < 0) enter_block(axis, inc[axis]);
    };
    // same case order as the reference; when no case applies the walk repeats the voxel (n still counts down)
    if (err_xy > 0 && err_xz > 0) {
        step(0);
        err_xy -= d2[1];
        err_xz -= d2[2];
    } else if (err_xy < 0 && err_yz > 0) {
        step(1);
        err_xy += d2[0];
        err_yz -= d2[2];
    } else if (err_yz < 0 && err_xz < 0) {
        step(2);
        err_xz += d2[0];
        err_yz += d2[1];
    } else if (err_xy == 0) {  // diagonal move in the xy plane: two voxel steps at once
        step(0);
        step(1);
        n -= 2;
    }
    --n;
    return valid;
}

// ----------------------------------------------------------------- BGKOctoMap
BGKOctoMap::BGKOctoMap() : BGKOctoMap(0.1f, 4, 1.0, 1.0, 0.3f, 0.7f, 1.0f, 1.0f, 1.0f) {}

BGKOctoMap::BGKOctoMap(float resolution_, unsigned short block_depth_, float sf2, float ell, float free_thresh,
                       float occupied_thresh, float var_thresh, float prior_A, float prior_B, int device)
    : BGKOctoMap(0, resolution_, block_depth_, sf2, ell, free_thresh, occupied_thresh, var_thresh, prior_A, prior_B,
                 nullptr, device) {}

GPOctoMap::GPOctoMap(float resolution_, unsigned short block_depth_, float sf2, float ell, float noise, float l,
                     float min_var, float max_var, float max_known_var, float free_thresh, float occupied_thresh,
                     int device)
    : BGKOctoMap(1, resolution_, block_depth_, sf2, ell, free_thresh, occupied_thresh, 0.0f, 0.0f, 0.0f,
                 [&] {
                     static thread_local GPParams g;
                     g = GPParams{noise, l, min_var, max_var, max_known_var, 0.1f, true};
                     return &g;
                 }(),
                 device) {}

BGKOctoMap::BGKOctoMap(int variant_, float resolution_, unsigned short block_depth_, float sf2, float ell,
                       float free_thresh, float occupied_thresh, float var_thresh, float prior_A, float prior_B,
                       const GPParams *gp, int device)
    : resolution(resolution_), block_size((float)pow(2, block_depth_ - 1) * resolution_), block_depth(block_depth_),
      ctx(nullptr), scan_flags(0), variant(variant_), train_max_n(0), train_sum_n2(0) {
    Block::resolution = resolution;
    Block::size = block_size;
    Block::key_loc_map = init_key_loc_map(resolution, block_depth);
    OcTree::max_depth = block_depth;
    OcTreeNode::sf2 = sf2;
    OcTreeNode::ell = ell;
    OcTreeNode::free_thresh = free_thresh;
    OcTreeNode::occupied_thresh = occupied_thresh;
    OcTreeNode::var_thresh = var_thresh;
    OcTreeNode::prior_A = prior_A;
    OcTreeNode::prior_B = prior_B;
    OcTreeNode::variant = variant == 3 ? 0 : variant;  // BGKLOctoMap shares the BGK node (bgkloctree_node.cpp)
    OcTreeNode::init_A = prior_A;
    OcTreeNode::init_B = prior_B;
    if (variant == 2) {  // src/bgklvoctomap/bgklvoctomap.cpp:60-61
        OcTreeNode::min_W = gp->min_W;
        OcTreeNode::original_size = gp->original_size;
    }
    if (variant == 1) {  // src/gpoctomap/gpoctomap.cpp:37-44
        OcTreeNode::noise = gp->noise;
        OcTreeNode::l = gp->l;
        OcTreeNode::min_ivar = 1.0f / gp->max_var;
        OcTreeNode::max_ivar = 1.0f / gp->min_var;
        OcTreeNode::min_known_ivar = 1.0f / gp->max_known_var;
        OcTreeNode::init_A = 0.0f;
        OcTreeNode::init_B = OcTreeNode::min_ivar;
    }

    capture_statics();

    la3dm_params &p = create_params;
    std::memset(&p, 0, sizeof(p));
    p.variant = variant;
    p.noise = OcTreeNode::noise;
    p.l = OcTreeNode::l;
    p.min_ivar = OcTreeNode::min_ivar;
    p.max_ivar = OcTreeNode::max_ivar;
    p.min_known_ivar = OcTreeNode::min_known_ivar;
    p.min_W = OcTreeNode::min_W;
    p.sf2 = sf2;
    p.ell = ell;
    p.free_thresh = free_thresh;
    p.occupied_thresh = occupied_thresh;
    p.var_thresh = var_thresh;
    p.prior_A = prior_A;
    p.prior_B = prior_B;
    p.device = device;
    create_context();
}

void BGKOctoMap::create_context() {
    la3dm_params &p = create_params;
    p.resolution = resolution;
    p.block_depth = block_depth;
    p.lut_xyz = &Block::key_loc_map[0].x();
    p.lut_count = (uint32_t)Block::key_loc_map.size();
    static_assert(sizeof(point3f) == 12, "LUT is handed to the device as packed xyz");
    if (p.device < 0) return;  // bookkeeping-only map (tests of the host logic): inserting throws
    int rc = la3dm_create(&p, &ctx);
    if (rc != LA3DM_OK)
        throw std::runtime_error(std::string("BGKOctoMap: GPU context creation failed: ") + la3dm_last_error(nullptr));
    // Device-resident by default where it exists (BGK and GP maps, block_depth <= 5): insert_pointcloud then runs
    // start to finish on the GPU (insert_training_data too).  The split prepare()/commit() form moves the map back to
    // the host-orchestrated mode on its own (ensure_host_mode).  LA3DM_DEVICE_RESIDENT=0 keeps the host mode.
    const char *env = getenv("LA3DM_DEVICE_RESIDENT");
    if (variant >= 0 && variant <= 3 && block_depth <= 5 && !(env && env[0] == '0')) {
        if (la3dm_devmap_create(ctx, &dmap) != LA3DM_OK) dmap = nullptr;
    }
}

void BGKOctoMap::set_resolution(float r) { reconfigure(r, block_depth); }
void BGKOctoMap::set_block_depth(unsigned short d) { reconfigure(resolution, d); }

// reference bgkoctomap.cpp:66-80: resolution / block_depth, Block::resolution, Block::size, Block::key_loc_map,
// OcTree::max_depth — on an empty map (see the header)
void BGKOctoMap::reconfigure(float r, unsigned short d) {
    bind();
    if (!(r > 0.0f) || !std::isfinite(r)) throw std::invalid_argument("set_resolution: resolution must be positive and finite");
    if (d < 1 || d > 6) throw std::invalid_argument("set_block_depth: block_depth must be in 1..6 (16-bit in-block keys)");
    bool empty = block_arr.empty();
    if (dmap != nullptr) {
        uint32_t nb = 0, npb = 0;
        if (la3dm_devmap_block_count(dmap, &nb, &npb) != LA3DM_OK || nb != 0) empty = false;
    }
    if (!empty)
        throw std::logic_error("set_resolution / set_block_depth: the map already holds blocks (the reference re-derives the "
                               "block size and the voxel LUT under them and corrupts the map; build a new map instead)");
    const bool was_resident = dmap != nullptr;
    la3dm_devmap_destroy(dmap);
    dmap = nullptr;
    la3dm_destroy(ctx);
    ctx = nullptr;
    resolution = r;
    block_depth = d;
    block_size = (float)pow(2, block_depth - 1) * resolution;
    Block::resolution = resolution;
    Block::size = block_size;
    Block::key_loc_map = init_key_loc_map(resolution, block_depth);
    OcTree::max_depth = block_depth;
    capture_statics();
    passes.clear();
    create_context();
    if (!was_resident && dmap != nullptr) {  // the map had been moved to the host-orchestrated mode: keep it there
        la3dm_devmap_destroy(dmap);
        dmap = nullptr;
    }
    // the new device map starts unsharded: a block-sharded map stays one
    if (dmap != nullptr && shard_cfg.world > 1 &&
        la3dm_devmap_set_shard(dmap, shard_cfg.rank, shard_cfg.world, shard_cfg.fn, shard_cfg.user) != LA3DM_OK)
        throw std::runtime_error(la3dm_last_error(ctx));
}

const BGKOctoMap *BGKOctoMap::bound = nullptr;

void BGKOctoMap::capture_statics() {
    mine.resolution = Block::resolution;
    mine.size = Block::size;
    mine.key_loc_map = Block::key_loc_map;
    mine.max_depth = OcTree::max_depth;
    mine.sf2 = OcTreeNode::sf2;
    mine.ell = OcTreeNode::ell;
    mine.free_thresh = OcTreeNode::free_thresh;
    mine.occupied_thresh = OcTreeNode::occupied_thresh;
    mine.var_thresh = OcTreeNode::var_thresh;
    mine.prior_A = OcTreeNode::prior_A;
    mine.prior_B = OcTreeNode::prior_B;
    mine.init_A = OcTreeNode::init_A;
    mine.init_B = OcTreeNode::init_B;
    mine.min_W = OcTreeNode::min_W;
    mine.noise = OcTreeNode::noise;
    mine.l = OcTreeNode::l;
    mine.min_ivar = OcTreeNode::min_ivar;
    mine.max_ivar = OcTreeNode::max_ivar;
    mine.min_known_ivar = OcTreeNode::min_known_ivar;
    mine.variant = OcTreeNode::variant;
    mine.original_size = OcTreeNode::original_size;
    bound = this;
}

// re-install this map's parameters into the process-global statics (no-op while this map is the bound one)
void BGKOctoMap::bind() const {
    if (bound == this) return;
    Block::resolution = mine.resolution;
    Block::size = mine.size;
    Block::key_loc_map = mine.key_loc_map;
    OcTree::max_depth = mine.max_depth;
    OcTreeNode::sf2 = mine.sf2;
    OcTreeNode::ell = mine.ell;
    OcTreeNode::free_thresh = mine.free_thresh;
    OcTreeNode::occupied_thresh = mine.occupied_thresh;
    OcTreeNode::var_thresh = mine.var_thresh;
    OcTreeNode::prior_A = mine.prior_A;
    OcTreeNode::prior_B = mine.prior_B;
    OcTreeNode::init_A = mine.init_A;
    OcTreeNode::init_B = mine.init_B;
    OcTreeNode::min_W = mine.min_W;
    OcTreeNode::noise = mine.noise;
    OcTreeNode::l = mine.l;
    OcTreeNode::min_ivar = mine.min_ivar;
    OcTreeNode::max_ivar = mine.max_ivar;
    OcTreeNode::min_known_ivar = mine.min_known_ivar;
    OcTreeNode::variant = mine.variant;
    OcTreeNode::original_size = mine.original_size;
    bound = this;
}

void BGKOctoMap::ensure_host_mode() {
    if (dmap != nullptr) set_device_resident(false);
}

BGKOctoMap::~BGKOctoMap() {
    bind();  // the blocks' destructors read the layer count
    for (auto &kv : block_arr) delete kv.second;
    bound = nullptr;
    la3dm_devmap_destroy(dmap);
    la3dm_destroy(ctx);
}

// ------------------------------------------------------------- device-resident mode
void BGKOctoMap::set_device_resident(bool on) {
    bind();
    if (on == (dmap != nullptr)) return;
    if (!on) {
        sync_mirror();
        la3dm_devmap_destroy(dmap);
        dmap = nullptr;
        return;
    }
    if (ctx == nullptr) throw std::runtime_error("BGKOctoMap::set_device_resident: the map has no GPU context");
    if (!block_arr.empty())
        throw std::runtime_error("BGKOctoMap::set_device_resident: a map that already holds host blocks stays host-orchestrated");
    if (la3dm_devmap_create(ctx, &dmap) != LA3DM_OK)
        throw std::runtime_error(std::string("BGKOctoMap::set_device_resident: ") + la3dm_last_error(ctx));
}

// Refresh the host mirror from the device pool: every block's nodes are overwritten with the device
// state (alpha, beta, state, classified); PRUNED children keep their collapsed parents' leaf role.
void BGKOctoMap::sync_mirror() const {
    bind();
    if (dmap == nullptr || !mirror_dirty) return;
    uint32_t nb = 0, npb = 0;
    if (la3dm_devmap_block_count(dmap, &nb, &npb) != LA3DM_OK)
        throw std::runtime_error(std::string("BGKOctoMap::sync_mirror: ") + la3dm_last_error(ctx));
    std::vector<int64_t> keys(nb);
    std::vector<float> A((size_t)nb * npb), B((size_t)nb * npb);
    std::vector<uint8_t> S((size_t)nb * npb);
    if (nb && la3dm_devmap_download(dmap, keys.data(), A.data(), B.data(), S.data()) != LA3DM_OK)
        throw std::runtime_error(std::string("BGKOctoMap::sync_mirror: ") + la3dm_last_error(ctx));
    for (uint32_t b = 0; b < nb; ++b) {
        auto it = block_arr.find(keys[b]);
        if (it == block_arr.end()) it = block_arr.emplace(keys[b], new Block(hash_key_to_block(keys[b]))).first;
        it->second->load_nodes(&A[(size_t)b * npb], &B[(size_t)b * npb], &S[(size_t)b * npb], npb);
    }
    mirror_dirty = false;
}

std::vector<float> BGKOctoMap::device_training_data() const {
    std::vector<float> out;
    if (dmap == nullptr) return out;
    uint32_t n = 0;
    la3dm_devmap_training_data(dmap, nullptr, 0, &n);
    out.resize(4 * (size_t)n);
    if (n && la3dm_devmap_training_data(dmap, out.data(), n, &n) != LA3DM_OK)
        throw std::runtime_error(std::string("BGKOctoMap::device_training_data: ") + la3dm_last_error(ctx));
    return out;
}

Block *BGKOctoMap::search(BlockHashKey key) const {
    sync_mirror();
    auto it = block_arr.find(key);
    return it == block_arr.end() ? nullptr : it->second;
}

OcTreeNode BGKOctoMap::search(point3f p) const {
    bind();
    Block *b = search(block_to_hash_key(p));
    return b == nullptr ? OcTreeNode() : OcTreeNode(b->search(p));
}

void BGKOctoMap::search_many(const float *xyz, size_t n, uint8_t *exists, float *A, float *B, uint8_t *state) const {
    bind();
    if (dmap != nullptr) {
        if (la3dm_devmap_search_host(dmap, xyz, (uint32_t)n, exists, A, B, state) != LA3DM_OK)
            throw std::runtime_error(std::string("BGKOctoMap::search_many: ") + la3dm_last_error(ctx));
        return;
    }
    for (size_t i = 0; i < n; ++i) {
        const point3f p(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        exists[i] = search(block_to_hash_key(p)) != nullptr;
        const OcTreeNode nd = search(p);
        A[i] = nd.m_A;
        B[i] = nd.m_B;
        state[i] = (uint8_t)nd.get_state();
    }
}

namespace {
// heightMapColor, include/common/markerarray_pub.h:21-76 (s = v = 1)
void height_map_color(double h, float *rgba) {
    h -= floor(h);
    h *= 6;
    const int i = (int)floor(h);
    double f = h - i;
    if (!(i & 1)) f = 1 - f;
    const double v = 1.0, m = 0.0, n = 1.0 - f;
    double r, g, b;
    switch (i) {
    case 6:
    case 0: r = v; g = n; b = m; break;
    case 1: r = n; g = v; b = m; break;
    case 2: r = m; g = v; b = n; break;
    case 3: r = m; g = n; b = v; break;
    case 4: r = n; g = m; b = v; break;
    case 5: r = v; g = m; b = n; break;
    default: r = 1; g = 0.5; b = 0.5; break;
    }
    rgba[0] = (float)r;
    rgba[1] = (float)g;
    rgba[2] = (float)b;
    rgba[3] = 1.0f;
}
}  // namespace

size_t BGKOctoMap::export_cells(State state, bool original_size, float min_z, float max_z, Cells &out) const {
    bind();
    if (state != State::OCCUPIED && state != State::FREE)
        throw std::runtime_error("BGKOctoMap::export_cells: state must be OCCUPIED or FREE");
    out.xyz_size.clear();
    out.rgba.clear();
    out.level.clear();
    if (min_z == max_z) {  // bgkoctomap_static_node.cpp:103-108
        point3f lo, hi;
        get_bbox(lo, hi);
        min_z = lo.z();
        max_z = hi.z();
    }
    if (dmap != nullptr) {
        uint64_t n = 0;
        if (la3dm_devmap_export_cells(dmap, (int)state, original_size ? 1 : 0, min_z, max_z, nullptr, nullptr, nullptr, 0, &n) != LA3DM_OK)
            throw std::runtime_error(std::string("BGKOctoMap::export_cells: ") + la3dm_last_error(ctx));
        out.xyz_size.resize(4 * n);
        out.rgba.resize(4 * n);
        out.level.resize(n);
        if (n && la3dm_devmap_export_cells(dmap, (int)state, original_size ? 1 : 0, min_z, max_z, out.xyz_size.data(), out.rgba.data(),
                                           out.level.data(), n, &n) != LA3DM_OK)
            throw std::runtime_error(std::string("BGKOctoMap::export_cells: ") + la3dm_last_error(ctx));
        return (size_t)n;
    }
    auto push = [&](float x, float y, float z, float size, const OcTreeNode &nd) {
        out.xyz_size.insert(out.xyz_size.end(), {x, y, z, size});
        out.level.push_back(size > 0 ? (int)log2(size / resolution) : 0);  // markerarray_pub.h:110-114
        float c[4] = {0.0f, 0.0f, 1.0f, 1.0f};                             // the marker's default colour
        if (state == State::OCCUPIED) {
            if (min_z < max_z) {
                const double h = (1.0 - std::min(std::max((z - min_z) / (max_z - min_z), 0.0f), 1.0f)) * 0.8;
                height_map_color(h, c);
            }
        } else {
            const float prob = nd.get_prob();
            if (prob < 0.5f) {
                c[0] = c[1] = c[2] = 0.8f;
            } else {
                height_map_color(std::min(2.0 - 2.0 * prob, 0.6), c);
            }
        }
        out.rgba.insert(out.rgba.end(), c, c + 4);
    };
    for (auto it = begin_leaf(); it != end_leaf(); ++it) {
        if (it.get_node().get_state() != state) continue;
        if (original_size) {
            const point3f p = it.get_loc();
            push(p.x(), p.y(), p.z(), it.get_size(), it.get_node());
        } else {
            for (const point3f &p : it.get_pruned_locs()) push(p.x(), p.y(), p.z(), resolution, it.get_node());
        }
    }
    return out.level.size();
}

void BGKOctoMap::get_bbox(point3f &lim_min, point3f &lim_max) const {
    bind();
    if (dmap != nullptr) {  // index box of the pool's keys; centre = (index - 524288) * size is monotone in the index
        lim_min = point3f(0, 0, 0);
        lim_max = point3f(0, 0, 0);
        uint32_t nb = 0, npb = 0;
        if (la3dm_devmap_block_count(dmap, &nb, &npb) != LA3DM_OK)
            throw std::runtime_error(std::string("BGKOctoMap::get_bbox: ") + la3dm_last_error(ctx));
        if (nb == 0) return;
        int32_t lo[3], hi[3];
        if (la3dm_devmap_key_bounds(dmap, lo, hi) != LA3DM_OK)
            throw std::runtime_error(std::string("BGKOctoMap::get_bbox: ") + la3dm_last_error(ctx));
        const BlockHashKey klo = ((int64_t)lo[0] << 40) | ((int64_t)lo[1] << 20) | (int64_t)lo[2];
        const BlockHashKey khi = ((int64_t)hi[0] << 40) | ((int64_t)hi[1] << 20) | (int64_t)hi[2];
        lim_min = hash_key_to_block(klo);
        lim_max = hash_key_to_block(khi);
        lim_min -= point3f(block_size, block_size, block_size) * 0.5;
        lim_max += point3f(block_size, block_size, block_size) * 0.5;
        return;
    }
    sync_mirror();
    lim_min = point3f(0, 0, 0);
    lim_max = point3f(0, 0, 0);
    bool first = true;
    for (auto &kv : block_arr) {
        const point3f c = kv.second->get_center();
        if (first) {
            lim_min = lim_max = c;
            first = false;
            continue;
        }
        for (unsigned a = 0; a < 3; ++a) {
            lim_min(a) = std::min(lim_min(a), c(a));
            lim_max(a) = std::max(lim_max(a), c(a));
        }
    }
    if (!first) {
        lim_min -= point3f(block_size, block_size, block_size) * 0.5;
        lim_max += point3f(block_size, block_size, block_size) * 0.5;
    }
}

BGKOctoMap::LeafIterator::LeafIterator(const BGKOctoMap *map)
    : block_it(map->block_arr.cbegin()), end_block(map->block_arr.cend()) {
    // skip nothing: every block has at least one leaf
    if (block_it != end_block) {
        leaf_it = block_it->second->begin_leaf();
        end_leaf = block_it->second->end_leaf();
    }
}

BGKOctoMap::LeafIterator &BGKOctoMap::LeafIterator::operator++() {
    ++leaf_it;
    if (leaf_it == end_leaf) {
        ++block_it;
        if (block_it != end_block) {
            leaf_it = block_it->second->begin_leaf();
            end_leaf = block_it->second->end_leaf();
        }
    }
    return *this;
}

std::vector<point3f> BGKOctoMap::LeafIterator::get_pruned_locs() const {
    // base-resolution voxel centres covered by a (possibly collapsed) leaf
    std::vector<point3f> out;
    const point3f c = get_loc();
    const float size = get_size();
    const float x0 = c.x() - size * 0.5 + Block::resolution * 0.5;
    const float y0 = c.y() - size * 0.5 + Block::resolution * 0.5;
    const float z0 = c.z() - size * 0.5 + Block::resolution * 0.5;
    const float x1 = c.x() + size * 0.5, y1 = c.y() + size * 0.5, z1 = c.z() + size * 0.5;
    for (float x = x0; x < x1; x += Block::resolution)
        for (float y = y0; y < y1; y += Block::resolution)
            for (float z = z0; z < z1; z += Block::resolution) out.emplace_back(x, y, z);
    return out;
}

// ---------------------------------------------------------------- front end (stage A)
namespace {

// Voxel-grid centroid filter with the semantics of pcl::VoxelGrid<PointXYZ> (PCL is not a
// dependency of this build): cell = floor(p * (1/leaf)) - floor(min * (1/leaf)), output cells in
// ascending linear index, centroid = fp32 sum in cloud order / (float)count.
bool grid_order_option(la3dm_ctx *ctx) {
    int v = 0;
    return ctx && la3dm_get_option(ctx, "grid_order", &v) == 0 && v == 1;
}

void voxel_grid_filter(const float *in, size_t n, float leaf, std::vector<float> &out, bool pcl_order = false) {
    out.clear();
    if (n == 0) return;
    const float inv = 1.0f / leaf;
    float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
    float mx[3] = {-mn[0], -mn[1], -mn[2]};
    for (size_t i = 0; i < n; ++i) {
        const float *p = in + 3 * i;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], p[a]);
            mx[a] = std::max(mx[a], p[a]);
        }
    }
    const int64_t ex = (int64_t)((mx[0] - mn[0]) * inv) + 1, ey = (int64_t)((mx[1] - mn[1]) * inv) + 1,
                  ez = (int64_t)((mx[2] - mn[2]) * inv) + 1;
    if (ex * ey * ez > (int64_t)std::numeric_limits<int32_t>::max()) {  // PCL gives the input back
        out.assign(in, in + 3 * n);
        return;
    }
    int lo[3], span[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = (int)std::floor(mn[a] * inv);
        span[a] = (int)std::floor(mx[a] * inv) - lo[a] + 1;
    }
    const int m1 = span[0], m2 = span[0] * span[1];
    const size_t ncell = (size_t)span[0] * span[1] * span[2];
    if (pcl_order) {
        // option "grid_order" 1 (verification mode, include/la3dm_hip.h): the order of the points inside a cell as pcl::VoxelGrid's own
        // std::sort leaves it — {cell, cloud index} pairs compared on the cell alone (src/bgkoctomap/bgkoctomap.cpp:419-431)
        std::vector<std::pair<unsigned, unsigned>> iv;
        iv.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            const float *p = in + 3 * i;
            if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
            const int c0 = (int)(std::floor(p[0] * inv) - (float)lo[0]);
            const int c1 = (int)(std::floor(p[1] * inv) - (float)lo[1]);
            const int c2 = (int)(std::floor(p[2] * inv) - (float)lo[2]);
            iv.emplace_back((unsigned)(c0 + c1 * m1 + c2 * m2), (unsigned)i);
        }
        struct PclLess {
            bool operator()(const std::pair<unsigned, unsigned> &a, const std::pair<unsigned, unsigned> &b) const { return a.first < b.first; }
        };
        std::sort(iv.begin(), iv.end(), PclLess());
        for (size_t i = 0; i < iv.size();) {
            size_t j = i;
            float sx = 0.f, sy = 0.f, sz = 0.f;
            for (; j < iv.size() && iv[j].first == iv[i].first; ++j) {
                const float *p = in + 3 * (size_t)iv[j].second;
                sx += p[0];
                sy += p[1];
                sz += p[2];
            }
            const float cnt = (float)(j - i);
            out.push_back(sx / cnt);
            out.push_back(sy / cnt);
            out.push_back(sz / cnt);
            i = j;
        }
        return;
    }
    if (ncell <= ((size_t)1 << 24)) {
        // dense accumulation: one pass in cloud order (the same per-cell summation order as sorting by
        // (cell, index)), then the occupied cells in ascending index
        struct Acc {
            float x, y, z;
            uint32_t n;
        };
        // zero pages on demand (calloc): only the touched part of the grid is ever paged in
        Acc *acc = static_cast<Acc *>(std::calloc(ncell, sizeof(Acc)));
        if (acc == nullptr) throw std::bad_alloc();
        // pass 1 (streaming, parallel): cell of every point.  pass 2: every thread of a small team owns a contiguous
        // range of cells and walks the whole cell list in cloud order, accumulating only its own cells — each cell
        // still sums its points in cloud order (one owner), and the owners' touched lists, each sorted, concatenate
        // to the ascending cell order.  The accumulation target is prefetched a few points ahead (the cloud arrives in
        // ray order: every access is a cache miss).
        std::vector<uint32_t> cells(n);
        const int team = n > 200000 ? kHostThreads : 1;
#pragma omp parallel for num_threads(team) schedule(static)
        for (long ii = 0; ii < (long)n; ++ii) {
            const size_t i = (size_t)ii;
            const float *p = in + 3 * i;
            if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) {
                cells[i] = 0xFFFFFFFFu;
                continue;
            }
            const int c0 = (int)(std::floor(p[0] * inv) - (float)lo[0]);
            const int c1 = (int)(std::floor(p[1] * inv) - (float)lo[1]);
            const int c2 = (int)(std::floor(p[2] * inv) - (float)lo[2]);
            cells[i] = (uint32_t)(c0 + c1 * m1 + c2 * m2);
        }
        std::vector<std::vector<uint32_t>> touched(team);
        std::vector<size_t> first(team + 1, 0);
#pragma omp parallel num_threads(team)
        {
            const int t = omp_get_thread_num(), nt = omp_get_num_threads();  // the runtime may grant fewer threads
            const uint32_t c_lo = (uint32_t)((uint64_t)ncell * t / nt), c_hi = (uint32_t)((uint64_t)ncell * (t + 1) / nt);
            std::vector<uint32_t> &mine = touched[t];
            constexpr size_t kAhead = 24;
            for (size_t i = 0; i < n; ++i) {
                if (i + kAhead < n) {
                    const uint32_t ca = cells[i + kAhead];
                    if (ca >= c_lo && ca < c_hi) __builtin_prefetch(&acc[ca], 1, 1);
                }
                const uint32_t cell = cells[i];
                if (cell < c_lo || cell >= c_hi) continue;  // (0xFFFFFFFF = non-finite point: owned by nobody)
                const float *p = in + 3 * i;
                Acc &a = acc[cell];
                if (a.n == 0) mine.push_back(cell);
                a.x += p[0];
                a.y += p[1];
                a.z += p[2];
                ++a.n;
            }
            std::sort(mine.begin(), mine.end());
        }
        for (int t = 0; t < team; ++t) first[t + 1] = first[t] + touched[t].size();
        out.resize(3 * first[team]);
#pragma omp parallel for num_threads(team) schedule(static)
        for (int t = 0; t < team; ++t) {
            size_t o = 3 * first[t];
            for (uint32_t cell : touched[t]) {
                const Acc &a = acc[cell];
                const float cnt = (float)a.n;
                out[o++] = a.x / cnt;
                out[o++] = a.y / cnt;
                out[o++] = a.z / cnt;
            }
        }
        std::free(acc);
        return;
    }
    std::vector<uint64_t> order;
    order.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const float *p = in + 3 * i;
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        const int c0 = (int)(std::floor(p[0] * inv) - (float)lo[0]);
        const int c1 = (int)(std::floor(p[1] * inv) - (float)lo[1]);
        const int c2 = (int)(std::floor(p[2] * inv) - (float)lo[2]);
        const uint32_t cell = (uint32_t)(c0 + c1 * m1 + c2 * m2);
        order.push_back(((uint64_t)cell << 32) | (uint32_t)i);
    }
    std::sort(order.begin(), order.end());
    size_t i = 0;
    while (i < order.size()) {
        const uint32_t cell = (uint32_t)(order[i] >> 32);
        float sx = 0.f, sy = 0.f, sz = 0.f;
        size_t j = i;
        for (; j < order.size() && (uint32_t)(order[j] >> 32) == cell; ++j) {
            const float *p = in + 3 * (uint32_t)order[j];
            sx += p[0];
            sy += p[1];
            sz += p[2];
        }
        const float cnt = (float)(j - i);
        out.push_back(sx / cnt);
        out.push_back(sy / cnt);
        out.push_back(sz / cnt);
        i = j;
    }
}

}  // namespace

void BGKOctoMap::get_training_data(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                                   float free_resolution, float max_range) {
    std::vector<float> packed(3 * n);
    for (size_t i = 0; i < n; ++i) {
        packed[3 * i] = xyz[stride * i];
        packed[3 * i + 1] = xyz[stride * i + 1];
        packed[3 * i + 2] = xyz[stride * i + 2];
    }
    std::vector<float> hits;
    const double tt0 = wall();
    const bool pcl_order = grid_order_option(ctx);
    if (ds_resolution < 0) hits.swap(packed); else voxel_grid_filter(packed.data(), n, ds_resolution, hits, pcl_order);
    const double tt1 = wall();

    xy.clear();
    std::vector<float> frees;
    const float x0 = origin.x(), y0 = origin.y(), z0 = origin.z();
    const size_t nh = hits.size() / 3;
    xy.reserve(4 * nh);
    frees.reserve(3 * nh * 8);
    size_t kept = 0;
    for (size_t i = 0; i < nh; ++i) {
        const float x = hits[3 * i], y = hits[3 * i + 1], z = hits[3 * i + 2];
        if (max_range > 0) {
            const double l = (point3f(x, y, z) - origin).norm();
            if (l > max_range) continue;
        }
        xy.push_back(x);
        xy.push_back(y);
        xy.push_back(z);
        xy.push_back(1.0f);
        ++kept;
        // free-space samples along the beam: the origin itself, then every free_resolution,
        // then one sample free_resolution short of the hit
        const float l = (float)sqrt((x - x0) * (x - x0) + (y - y0) * (y - y0) + (z - z0) * (z - z0));
        const float nx = (x - x0) / l, ny = (y - y0) / l, nz = (z - z0) / l;
        size_t w = frees.size();
        const size_t room = 3 * ((l < 1.0e9f ? (size_t)(l / free_resolution) : 0) + 4);  // (false for a NaN range too)
        frees.resize(w + room);
        float *f = frees.data();
        f[w++] = x0;
        f[w++] = y0;
        f[w++] = z0;
        for (float d = free_resolution; d < l; d += free_resolution) {
            if (w + 6 > frees.size()) {  // (float stepping can take one more step than l / free_resolution)
                frees.resize(frees.size() + 64);
                f = frees.data();
            }
            f[w++] = x0 + nx * d;
            f[w++] = y0 + ny * d;
            f[w++] = z0 + nz * d;
        }
        if (l > free_resolution) {
            const float d = l - free_resolution;
            f[w++] = x0 + nx * d;
            f[w++] = y0 + ny * d;
            f[w++] = z0 + nz * d;
        }
        frees.resize(w);
    }
    std::vector<float> sampled;
    const double tt2 = wall();
    if (ds_resolution < 0) sampled.swap(frees); else voxel_grid_filter(frees.data(), frees.size() / 3, ds_resolution, sampled, pcl_order);
    if (getenv("LA3DM_TIMING"))
        fprintf(stderr, "[la3dm] front end: grid(hits) %.4f beam %.4f grid(frees, %zu pts) %.4f\n", tt1 - tt0, tt2 - tt1,
                frees.size() / 3, wall() - tt2);
    const size_t nf = sampled.size() / 3;
    xy.reserve(xy.size() + 4 * nf);
    const float free_label = variant == 1 ? -1.0f : 0.0f;  // bgkoctomap.cpp:415 / gpoctomap.cpp:399
    for (size_t i = 0; i < nf; ++i) xy.insert(xy.end(), {sampled[3 * i], sampled[3 * i + 1], sampled[3 * i + 2], free_label});
    stats.n_hits = kept;
    stats.n_frees = nf;
}

// BGKLOctoMap front end (src/bgkloctomap/bgkloctomap.cpp:300-343, beam_sample :359-381): hits are re-projected as
// origin + n * l, free samples step down from l - free_resolution while d > 0 and are not voxel-filtered, every sample
// of a beam (the origin sample first) remembers its beam; the beam itself is origin -> origin + n * (l - free_resolution).
void BGKOctoMap::get_training_data_l(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                                     float free_resolution, float max_range) {
    std::vector<float> packed(3 * n);
    for (size_t i = 0; i < n; ++i) {
        packed[3 * i] = xyz[stride * i];
        packed[3 * i + 1] = xyz[stride * i + 1];
        packed[3 * i + 2] = xyz[stride * i + 2];
    }
    std::vector<float> hits;
    if (ds_resolution < 0) hits.swap(packed); else voxel_grid_filter(packed.data(), n, ds_resolution, hits, grid_order_option(ctx));
    const float x0 = origin.x(), y0 = origin.y(), z0 = origin.z();
    const size_t nh = hits.size() / 3;
    // pass 1: range gate and sample count per hit (the float-stepped while loop, kept verbatim); pass 2: fill
    std::vector<uint32_t> first(nh + 1, 0u);  // first xy entry of hit i; kept hits only contribute
    std::vector<uint8_t> kept(nh, 0);
    auto end_point = [&](size_t i, float &ex, float &ey, float &ez, float &nx, float &ny, float &nz, float &l) {
        const float x = hits[3 * i], y = hits[3 * i + 1], z = hits[3 * i + 2];
        l = (float)sqrt((x - x0) * (x - x0) + (y - y0) * (y - y0) + (z - z0) * (z - z0));
        nx = (x - x0) / l;
        ny = (y - y0) / l;
        nz = (z - z0) / l;
        ex = x0 + nx * l;
        ey = y0 + ny * l;
        ez = z0 + nz * l;
    };
#pragma omp parallel for num_threads(kHostThreads) schedule(static)
    for (long i = 0; i < (long)nh; ++i) {
        const float x = hits[3 * i], y = hits[3 * i + 1], z = hits[3 * i + 2];
        if (max_range > 0) {
            const double lr = (point3f(x, y, z) - origin).norm();
            if (lr > max_range) continue;
        }
        kept[i] = 1;
        float ex, ey, ez, nx, ny, nz, l;
        end_point((size_t)i, ex, ey, ez, nx, ny, nz, l);
        const float l2 = (float)sqrt((ex - x0) * (ex - x0) + (ey - y0) * (ey - y0) + (ez - z0) * (ez - z0));
        uint32_t c = 2;  // the re-projected hit and the origin sample
        for (float d = l2 - free_resolution; d > 0.0; d -= free_resolution) ++c;
        first[i + 1] = c;
    }
    std::vector<int32_t> beam(nh, -1);
    int32_t idx = 0;
    for (size_t i = 0; i < nh; ++i) {
        first[i + 1] += first[i];
        if (kept[i]) beam[i] = idx++;
    }
    xy.resize(4 * (size_t)first[nh]);
    l_ray_idx.resize(first[nh]);
    l_rays.resize(6 * (size_t)idx);
#pragma omp parallel for num_threads(kHostThreads) schedule(static)
    for (long i = 0; i < (long)nh; ++i) {
        if (!kept[i]) continue;
        float ex, ey, ez, nx, ny, nz, l;
        end_point((size_t)i, ex, ey, ez, nx, ny, nz, l);
        const int32_t id = beam[i];
        size_t w = first[i];
        auto put = [&](float a, float b, float c, float lab, int32_t ray) {
            float *o = &xy[4 * w];
            o[0] = a;
            o[1] = b;
            o[2] = c;
            o[3] = lab;
            l_ray_idx[w++] = ray;
        };
        put(ex, ey, ez, 1.0f, -1);
        put(x0, y0, z0, 0.0f, id);
        {   // beam_sample from the re-projected end point
            const float l2 = (float)sqrt((ex - x0) * (ex - x0) + (ey - y0) * (ey - y0) + (ez - z0) * (ez - z0));
            const float mx = (ex - x0) / l2, my = (ey - y0) / l2, mz = (ez - z0) / l2;
            float d = l2 - free_resolution;
            while (d > 0.0) {
                put(x0 + mx * d, y0 + my * d, z0 + mz * d, 0.0f, id);
                d -= free_resolution;
            }
        }
        l = l - free_resolution;
        const float ray[6] = {x0, y0, z0, x0 + nx * l, y0 + ny * l, z0 + nz * l};
        std::memcpy(&l_rays[6 * (size_t)id], ray, sizeof(ray));
    }
    stats.n_hits = (uint64_t)idx;
    stats.n_frees = xy.size() / 4 - (uint64_t)idx;
}

// Training rows of every block (bgkloctomap.cpp:141-170): members in CSR order; a hit becomes a degenerate segment with
// label 1, a free sample contributes its beam once per block (at the position of the beam's first sample) with label 0.
void BGKOctoMap::build_rows_l() {
    // Members of a block are in ascending source order and the samples of one beam are consecutive in `xy`, so the
    // entries of a beam are adjacent inside a block: "once per block" = "differs from the previous entry's beam".
    const long nblk = (long)train_off.size() - 1;
    rows_off.assign((size_t)std::max(nblk, 0L) + 1, 0u);
    auto emits = [&](uint32_t k, uint32_t k0) {
        const int32_t r = l_ray_idx[train_src[k]];
        return r < 0 || k == k0 || l_ray_idx[train_src[k - 1]] != r;
    };
#pragma omp parallel for num_threads(kHostThreads) schedule(dynamic, 256)
    for (long b = 0; b < nblk; ++b) {
        uint32_t c = 0;
        for (uint32_t k = train_off[b]; k < train_off[b + 1]; ++k) c += emits(k, train_off[b]) ? 1u : 0u;
        rows_off[b + 1] = c;
    }
    for (long b = 0; b < nblk; ++b) rows_off[b + 1] += rows_off[b];
    train_rows.resize(8 * (size_t)rows_off[(size_t)std::max(nblk, 0L)]);
#pragma omp parallel for num_threads(kHostThreads) schedule(dynamic, 256)
    for (long b = 0; b < nblk; ++b) {
        float *o = train_rows.data() + 8 * (size_t)rows_off[b];
        for (uint32_t k = train_off[b]; k < train_off[b + 1]; ++k) {
            if (!emits(k, train_off[b])) continue;
            const uint32_t src = train_src[k];
            const int32_t r = l_ray_idx[src];
            if (r < 0) {
                const float *p = &xy[4 * (size_t)src];
                const float row[8] = {p[0], p[1], p[2], p[0], p[1], p[2], 1.0f, 0.0f};
                std::memcpy(o, row, sizeof(row));
            } else {
                const float *q = &l_rays[6 * (size_t)r];
                const float row[8] = {q[0], q[1], q[2], q[3], q[4], q[5], 0.0f, 0.0f};
                std::memcpy(o, row, sizeof(row));
            }
            o += 8;
        }
    }
    // work counters in rows (what the oracle counts for this variant)
    stats.train_reads = stats.pair_evals = 0;
    for (const Pass &ps : passes)
        for (size_t t = 0; t < ps.blocks.size(); ++t) {
            uint64_t nr = 0;
            for (int q = 0; q < 7; ++q) {
                const int32_t tb = ps.nbr[7 * t + q];
                if (tb >= 0) nr += rows_off[tb + 1] - rows_off[tb];
            }
            stats.train_reads += nr;
            stats.pair_evals += nr * (ps.leaf_off[t + 1] - ps.leaf_off[t]);
        }
}

// ------------------------------------------------- partition + pack (stages B..E host part)
namespace {
struct AxisCand {
    int64_t idx[3];
    int n;
};
// block indices (with the +524288 bias) whose CLOSED fp32 box [c-h, c+h] holds v on one axis
inline AxisCand axis_candidates(float v, float size, float h) {
    AxisCand r;
    r.n = 0;
    const int64_t i0 = int64_t(v / (double)size + 524288.5);
    for (int64_t i = i0 - 1; i <= i0 + 1; ++i) {
        const float c = (i - 524288) * size;
        if (c - h <= v && v <= c + h) r.idx[r.n++] = i;
    }
    return r;
}
}  // namespace

bool BGKOctoMap::partition_and_pack(bool ungated) {
    scan_flags = ungated ? LA3DM_SCAN_UPDATE_UNGATED : 0u;
    passes.clear();
    prune_list.clear();
    train_xyzy.clear();
    train_src.clear();
    train_off.assign(1, 0u);
    const size_t npts = xy.size() / 4;
    if (npts == 0) return false;
    const double t0 = wall();
    const float bs = block_size;
    const float h = bs / 2.0f;

    // bounding box of the training set, candidate block list (float-stepped, repeats kept)
    float lo[3] = {xy[0], xy[1], xy[2]}, hi[3] = {xy[0], xy[1], xy[2]};
    {
        float l0 = lo[0], l1 = lo[1], l2 = lo[2], h0 = hi[0], h1 = hi[1], h2 = hi[2];
#pragma omp parallel for num_threads(npts > 100000 ? kHostThreads : 1) schedule(static) reduction(min : l0, l1, l2) reduction(max : h0, h1, h2)
        for (long i = 1; i < (long)npts; ++i) {
            const float *p = &xy[4 * (size_t)i];
            l0 = std::min(l0, p[0]);
            l1 = std::min(l1, p[1]);
            l2 = std::min(l2, p[2]);
            h0 = std::max(h0, p[0]);
            h1 = std::max(h1, p[1]);
            h2 = std::max(h2, p[2]);
        }
        lo[0] = l0; lo[1] = l1; lo[2] = l2;
        hi[0] = h0; hi[1] = h1; hi[2] = h2;
        // The reference's reduction uses `<`, which a NaN never wins: NaN coordinates are ignored, except in the first
        // point, which seeds the reduction and never loses — that axis' limits are NaN and no candidate block is made.
        for (int a = 0; a < 3; ++a)
            if (xy[a] != xy[a]) lo[a] = hi[a] = xy[a];
    }
    std::vector<BlockHashKey> bbox_keys;
    for (float x = lo[0] - bs; x <= hi[0] + 2 * bs; x += bs)
        for (float y = lo[1] - bs; y <= hi[1] + 2 * bs; y += bs)
            for (float z = lo[2] - bs; z <= hi[2] + 2 * bs; z += bs) bbox_keys.push_back(block_to_hash_key(x, y, z));
    stats.n_bbox_blocks = bbox_keys.size();
    std::unordered_map<BlockHashKey, int32_t> in_bbox;  // key -> training-block index (or -1)
    in_bbox.reserve(bbox_keys.size() * 2);
    for (BlockHashKey k : bbox_keys) in_bbox.emplace(k, -1);

    // membership: every (block, point) pair with the point in the block's closed box
    struct Member {
        BlockHashKey key;
        uint32_t pt;
    };
    std::vector<Member> members;
    {
        // contiguous point ranges per thread, concatenated in range order: the list is the serial loop's list
        const int team = npts > 100000 ? kHostThreads : 1;
        std::vector<std::vector<Member>> part((size_t)team);
#pragma omp parallel for num_threads(team) schedule(static, 1)
        for (int t = 0; t < team; ++t) {
            const size_t i0 = npts * (size_t)t / (size_t)team, i1 = npts * (size_t)(t + 1) / (size_t)team;
            std::vector<Member> &out = part[(size_t)t];
            out.reserve((i1 - i0) + (i1 - i0) / 8);
            for (size_t i = i0; i < i1; ++i) {
                const AxisCand ax = axis_candidates(xy[4 * i], bs, h), ay = axis_candidates(xy[4 * i + 1], bs, h),
                               az = axis_candidates(xy[4 * i + 2], bs, h);
                for (int a = 0; a < ax.n; ++a)
                    for (int b = 0; b < ay.n; ++b)
                        for (int c = 0; c < az.n; ++c)
                            out.push_back(Member{(ax.idx[a] << 40) | (ay.idx[b] << 20) | az.idx[c], (uint32_t)i});
            }
        }
        std::vector<size_t> off((size_t)team + 1, 0);
        for (int t = 0; t < team; ++t) off[(size_t)t + 1] = off[(size_t)t] + part[(size_t)t].size();
        members.resize(off[(size_t)team]);
#pragma omp parallel for num_threads(team) schedule(static, 1)
        for (int t = 0; t < team; ++t)
            if (!part[(size_t)t].empty())
                std::memcpy(members.data() + off[(size_t)t], part[(size_t)t].data(), part[(size_t)t].size() * sizeof(Member));
    }
    // group by block; inside a block keep ascending point index.  (libstdc++ parallel mode, a small team: the sort is
    // the largest single item of the host partition; a stable sort has one result whatever the thread count.)
    if (members.size() > 100000)
        __gnu_parallel::stable_sort(members.begin(), members.end(), [](const Member &a, const Member &b) { return a.key < b.key; },
                                    __gnu_parallel::default_parallel_tag(kHostThreads));
    else
        std::stable_sort(members.begin(), members.end(), [](const Member &a, const Member &b) { return a.key < b.key; });

    // "geo" blocks = every block that geometrically holds points (what an R-tree query sees);
    // trained blocks = geo blocks that are also in the candidate list.
    std::unordered_map<BlockHashKey, char> geo;
    uint32_t n_train_blk = 0;
    {
        std::vector<size_t> start;  // first member of every block group (+ end)
        for (size_t i = 0; i < members.size(); ++i)
            if (i == 0 || members[i].key != members[i - 1].key) start.push_back(i);
        start.push_back(members.size());
        geo.reserve(start.size() * 2 + 16);
        std::vector<size_t> tgroup;  // groups that are training blocks, in key order
        for (size_t g = 0; g + 1 < start.size(); ++g) {
            const BlockHashKey key = members[start[g]].key;
            geo.emplace(key, 1);
            auto it = in_bbox.find(key);
            if (it == in_bbox.end()) continue;
            it->second = (int32_t)n_train_blk++;
            tgroup.push_back(g);
            train_off.push_back(train_off.back() + (uint32_t)(start[g + 1] - start[g]));
        }
        train_xyzy.resize(4 * (size_t)train_off.back());
        if (variant == 3) train_src.resize(train_off.back());
#pragma omp parallel for num_threads(kHostThreads) schedule(dynamic, 256)
        for (long b = 0; b < (long)tgroup.size(); ++b) {
            const size_t g = tgroup[(size_t)b];
            size_t w = train_off[(size_t)b];
            for (size_t k = start[g]; k < start[g + 1]; ++k, ++w) {
                std::memcpy(&train_xyzy[4 * w], &xy[4 * (size_t)members[k].pt], 4 * sizeof(float));
                if (variant == 3) train_src[w] = members[k].pt;
            }
        }
    }
    stats.n_train_blocks = n_train_blk;
    train_max_n = 0;
    train_sum_n2 = 0;
    for (size_t b = 0; b + 1 < train_off.size(); ++b) {
        const uint64_t nb = train_off[b + 1] - train_off[b];
        train_max_n = std::max<uint32_t>(train_max_n, (uint32_t)nb);
        train_sum_n2 += nb * nb;
    }

    // test blocks: candidate blocks whose extended block holds any point; list order kept
    std::unordered_map<BlockHashKey, uint32_t> times_seen;
    std::vector<std::pair<BlockHashKey, uint32_t>> test;  // (key, occurrence number)
    for (BlockHashKey k : bbox_keys) {
        const ExtendedBlock e = get_extended_block(k);
        bool any = false;
        for (int q = 0; q < 7 && !any; ++q) any = geo.find(e[q]) != geo.end();
        if (!any) continue;
        test.emplace_back(k, times_seen[k]++);
        prune_list.push_back(k);
    }
    stats.n_test_blocks = test.size();
    // Device-side load balance: hand the GPU the heaviest test blocks first (longest-
    // processing-time order, weight = training points in the 7-neighbourhood).  The blocks are
    // independent, so the order only changes the layout of the packed arrays, not the results.
    {
        auto weight = [&](BlockHashKey k) {
            const ExtendedBlock e = get_extended_block(k);
            uint32_t w = 0;
            for (int q = 0; q < 7; ++q) {
                auto it = in_bbox.find(e[q]);
                if (it != in_bbox.end() && it->second >= 0) w += train_off[it->second + 1] - train_off[it->second];
            }
            return w;
        };
        std::vector<std::pair<uint32_t, uint32_t>> order(test.size());  // (weight, position)
        for (size_t i = 0; i < test.size(); ++i) order[i] = {weight(test[i].first), (uint32_t)i};
        std::stable_sort(order.begin(), order.end(),
                         [](const std::pair<uint32_t, uint32_t> &a, const std::pair<uint32_t, uint32_t> &b) { return a.first > b.first; });
        std::vector<std::pair<BlockHashKey, uint32_t>> sorted(test.size());
        for (size_t i = 0; i < test.size(); ++i) sorted[i] = test[order[i].second];
        test.swap(sorted);
    }
    const double t1 = wall();
    stats.t_partition = t1 - t0;

    uint32_t max_occ = 0;
    for (auto &t : test) max_occ = std::max(max_occ, t.second);
    passes.resize(test.empty() ? 0 : max_occ + 1);
    stats.voxel_updates = stats.train_reads = stats.pair_evals = 0;
    // find or create the blocks (bgkoctomap.cpp:298-305): look-ups and insertions in list order on one thread, the
    // constructors of the missing ones (a node slab each) on the team
    const long ntest = (long)test.size();
    std::vector<Block *> tblk((size_t)ntest, nullptr);
    {
        std::vector<long> missing;
        std::unordered_map<BlockHashKey, long> first_missing;
        for (long t = 0; t < ntest; ++t) {
            auto bit = block_arr.find(test[(size_t)t].first);
            if (bit != block_arr.end()) tblk[(size_t)t] = bit->second;
            else if (first_missing.emplace(test[(size_t)t].first, t).second) missing.push_back(t);
        }
#pragma omp parallel for num_threads(kHostThreads) schedule(static)
        for (long k = 0; k < (long)missing.size(); ++k) {
            const long t = missing[(size_t)k];
            tblk[(size_t)t] = new Block(hash_key_to_block(test[(size_t)t].first));
        }
        for (long t : missing) block_arr.emplace(test[(size_t)t].first, tblk[(size_t)t]);
        for (long t = 0; t < ntest; ++t)
            if (tblk[(size_t)t] == nullptr) tblk[(size_t)t] = tblk[(size_t)first_missing[test[(size_t)t].first]];
    }
    // position of every test block inside its pass, then neighbour table / centre / leaf count on the team
    std::vector<uint32_t> pos((size_t)ntest), nleaf((size_t)ntest);
    for (long t = 0; t < ntest; ++t) {
        Pass &ps = passes[test[(size_t)t].second];
        pos[(size_t)t] = (uint32_t)ps.blocks.size();
        ps.keys.push_back(test[(size_t)t].first);
        ps.blocks.push_back(tblk[(size_t)t]);
    }
    for (Pass &ps : passes) {
        ps.center.resize(3 * ps.blocks.size());
        ps.nbr.resize(7 * ps.blocks.size());
        ps.leaf_off.assign(ps.blocks.size() + 1, 0u);
    }
    uint64_t sum_reads = 0, sum_pairs = 0, sum_leaves = 0;
#pragma omp parallel num_threads(kHostThreads) reduction(+ : sum_reads, sum_pairs, sum_leaves)
    {
        std::vector<uint32_t> scratch;
#pragma omp for schedule(static)
        for (long t = 0; t < ntest; ++t) {
            Pass &ps = passes[test[(size_t)t].second];
            const uint32_t b = pos[(size_t)t];
            Block *blk = tblk[(size_t)t];
            const point3f c = blk->get_center();
            ps.center[3 * (size_t)b] = c.x();
            ps.center[3 * (size_t)b + 1] = c.y();
            ps.center[3 * (size_t)b + 2] = c.z();
            const ExtendedBlock e = blk->get_extended_block();
            uint64_t npts_nb = 0;
            for (int q = 0; q < 7; ++q) {
                auto it = in_bbox.find(e[q]);
                const int32_t tb = it == in_bbox.end() ? -1 : it->second;
                ps.nbr[7 * (size_t)b + q] = tb;
                if (tb >= 0) npts_nb += train_off[tb + 1] - train_off[tb];
            }
            scratch.clear();
            blk->collect_leaves(scratch);
            nleaf[(size_t)t] = (uint32_t)scratch.size();
            ps.leaf_off[(size_t)b + 1] = (uint32_t)scratch.size();
            sum_leaves += scratch.size();
            sum_reads += npts_nb;
            sum_pairs += npts_nb * scratch.size();
        }
    }
    stats.voxel_updates = sum_leaves;
    stats.train_reads = sum_reads;
    stats.pair_evals = sum_pairs;
    for (Pass &ps : passes) {
        for (size_t b = 0; b < ps.blocks.size(); ++b) ps.leaf_off[b + 1] += ps.leaf_off[b];
        ps.leaf_key.resize(ps.leaf_off.back());
    }
#pragma omp parallel num_threads(kHostThreads)
    {
        std::vector<uint32_t> scratch;
#pragma omp for schedule(static)
        for (long t = 0; t < ntest; ++t) {
            Pass &ps = passes[test[(size_t)t].second];
            scratch.clear();
            tblk[(size_t)t]->collect_leaves(scratch);
            std::memcpy(ps.leaf_key.data() + ps.leaf_off[pos[(size_t)t]], scratch.data(), scratch.size() * sizeof(uint32_t));
        }
    }
    for (Pass &ps : passes) {
        const size_t nl = ps.leaf_key.size();
        ps.alpha.resize(nl);
        ps.beta.resize(nl);
        ps.state.assign(nl, 0);
#pragma omp parallel for num_threads(kHostThreads) schedule(static)
        for (long b = 0; b < (long)ps.blocks.size(); ++b)
            for (size_t l = ps.leaf_off[b]; l < ps.leaf_off[b + 1]; ++l) {
                const OcTreeNode &nd = (*ps.blocks[b])[(OcTreeHashKey)ps.leaf_key[l]];
                ps.alpha[l] = nd.m_A;
                ps.beta[l] = nd.m_B;
            }
    }
    stats.t_pack = wall() - t1;
    return !passes.empty();
}

int BGKOctoMap::run_scan(la3dm_bgk_scan *s, la3dm_bgk_counters *c) {
    if (variant == 3) return la3dm_bgkl_scan_host(ctx, s, c);
    return variant == 1 ? la3dm_gp_scan_host(ctx, s, c) : la3dm_bgk_scan_host(ctx, s, c);
}

la3dm_bgk_scan BGKOctoMap::packed(size_t pass) {
    bind();
    la3dm_bgk_scan s;
    std::memset(&s, 0, sizeof(s));
    Pass &ps = passes.at(pass);
    s.train_xyzy = train_xyzy.data();
    s.train_off = train_off.data();
    s.n_train_pts = (uint32_t)(train_xyzy.size() / 4);
    s.n_train_blk = (uint32_t)(train_off.size() - 1);
    s.nbr = ps.nbr.data();
    s.blk_center = ps.center.data();
    s.leaf_off = ps.leaf_off.data();
    s.n_test_blk = (uint32_t)ps.blocks.size();
    s.n_leaf = (uint32_t)ps.leaf_key.size();
    s.leaf_key = ps.leaf_key.data();
    s.alpha = ps.alpha.data();
    s.beta = ps.beta.data();
    s.state = ps.state.data();
    s.flags = scan_flags;
    if ((uint64_t)s.n_leaf == ((uint64_t)s.n_test_blk << (3 * (block_depth - 1)))) s.flags |= LA3DM_SCAN_FULL_BLOCKS;  // no test block of this pass is pruned
    s.train_max_n = train_max_n;
    s.train_sum_n2 = train_sum_n2;
    if (variant == 3) {  // BGKLOctoMap: segment rows (8 floats) instead of points
        s.train_xyzy = train_rows.data();
        s.train_off = rows_off.data();
        s.n_train_pts = (uint32_t)(train_rows.size() / 8);
    }
    return s;
}

void BGKOctoMap::refresh_pass(size_t p) {
    Pass &ps = passes[p];
#pragma omp parallel for num_threads(kHostThreads) schedule(static)
    for (long b = 0; b < (long)ps.blocks.size(); ++b)
        for (size_t l = ps.leaf_off[b]; l < ps.leaf_off[b + 1]; ++l) {
            const OcTreeNode &nd = (*ps.blocks[b])[(OcTreeHashKey)ps.leaf_key[l]];
            ps.alpha[l] = nd.m_A;
            ps.beta[l] = nd.m_B;
        }
}

void BGKOctoMap::write_nodes(size_t p) {
    Pass &ps = passes[p];
#pragma omp parallel for num_threads(kHostThreads) schedule(static)
    for (long b = 0; b < (long)ps.blocks.size(); ++b)
        for (size_t l = ps.leaf_off[b]; l < ps.leaf_off[b + 1]; ++l) {
            const uint8_t st = ps.state[l];
            if (!(st & LA3DM_LEAF_UPDATED)) continue;
            OcTreeNode &nd = (*ps.blocks[b])[(OcTreeHashKey)ps.leaf_key[l]];
            nd.classified = true;
            nd.m_A = ps.alpha[l];
            nd.m_B = ps.beta[l];
            nd.state = (State)(st & 3u);
        }
}

// Pass 0 (all distinct test blocks) has been run by the caller; repeated keys of the
// candidate list (a float-stepping artefact, normally none) are replayed here one pass
// after the other so that they see the previous pass's posterior, as the serial reference does.
void BGKOctoMap::commit() {
    bind();
    const double t0 = wall();
    if (!passes.empty()) write_nodes(0);
    for (size_t p = 1; p < passes.size(); ++p) {
        refresh_pass(p);
        la3dm_bgk_scan s = packed(p);
        if (ctx == nullptr) throw std::runtime_error("BGKOctoMap::commit: no device context");
        if (run_scan(&s, nullptr) != LA3DM_OK)
            throw std::runtime_error(std::string("BGKOctoMap::commit: ") + la3dm_last_error(ctx));
        write_nodes(p);
    }
    const double t1 = wall();
    stats.t_commit = t1 - t0;
    // the blocks of pass 0 are the distinct test blocks (prune_list only repeats some of them; prune is idempotent)
    if (!passes.empty()) {
        std::vector<Block *> &blks = passes[0].blocks;
#pragma omp parallel for num_threads(kHostThreads) schedule(static)
        for (long b = 0; b < (long)blks.size(); ++b) blks[b]->prune();
    }
    stats.t_prune = wall() - t1;
}

void BGKOctoMap::set_shard(uint32_t rank, uint32_t world, la3dm_allgatherv_fn fn, void *user) {
    if (dmap == nullptr) throw std::runtime_error("set_shard: the map is not in device-resident mode");
    if (la3dm_devmap_set_shard(dmap, rank, world, fn, user) != LA3DM_OK) throw std::runtime_error(la3dm_last_error(ctx));
    shard_cfg = {rank, world, fn, user};   // (reconfigure() rebuilds the device map and applies it again)
}

bool BGKOctoMap::prepare(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                         float free_res, float max_range) {
    bind();
    ensure_host_mode();
    stats = ScanStats();
    const double t0 = wall();
    if (variant == 3) {
        get_training_data_l(xyz, n, stride, origin, ds_resolution, free_res, max_range);
        stats.t_frontend = wall() - t0;
        const bool work = partition_and_pack(false);
        build_rows_l();
        return work;
    }
    get_training_data(xyz, n, stride, origin, ds_resolution, free_res, max_range);
    stats.t_frontend = wall() - t0;
    const bool work = partition_and_pack(false);
    scan_flags |= LA3DM_SCAN_LABELS_01;  // hits are labelled 1.0f, free samples 0.0f
    return work;
}

bool BGKOctoMap::prepare_training_data(const float *xyzy, size_t n, bool ungated) {
    bind();
    ensure_host_mode();
    stats = ScanStats();
    xy.assign(xyzy, xyzy + 4 * n);
    for (size_t i = 0; i < n; ++i) (xyzy[4 * i + 3] > 0.5f ? stats.n_hits : stats.n_frees)++;
    return partition_and_pack(ungated);
}

void BGKOctoMap::take_device_stats(const la3dm_devmap_stats &ds) {
    stats = ScanStats();
    stats.n_hits = ds.n_hits;
    stats.n_frees = ds.n_frees;
    stats.n_bbox_blocks = ds.n_bbox_blocks;
    stats.n_train_blocks = ds.n_train_blocks;
    stats.n_test_blocks = ds.n_test_blocks;
    stats.voxel_updates = ds.voxel_updates;
    stats.train_reads = ds.train_reads;
    stats.pair_evals = ds.pair_evals;
    stats.t_frontend = ds.t_frontend;
    stats.t_partition = ds.t_partition;
    stats.t_pack = ds.t_pack;
    stats.t_device = ds.t_kernel;
    stats.t_commit = ds.t_commit;
    stats.t_gather = ds.t_gather;
}

void BGKOctoMap::insert_pointcloud(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                                   float free_res, float max_range) {
    bind();
    const double t0 = wall();
    if (ctx == nullptr) throw std::runtime_error("BGKOctoMap::insert_pointcloud: no device context (there is no CPU path)");
    if (dmap != nullptr) {  // device-resident mode: the whole scan runs on the GPU
        const float o[3] = {origin.x(), origin.y(), origin.z()};
        la3dm_devmap_stats ds;
        if (la3dm_devmap_insert_pointcloud_host(dmap, xyz, (uint32_t)n, (uint32_t)stride, o, ds_resolution, free_res, max_range,
                                                &ds) != LA3DM_OK)
            throw std::runtime_error(std::string("BGKOctoMap::insert_pointcloud: ") + la3dm_last_error(ctx));
        take_device_stats(ds);
        stats.t_total = wall() - t0;
        mirror_dirty = true;
        return;
    }
    if (!prepare(xyz, n, stride, origin, ds_resolution, free_res, max_range)) return;
    const double t1 = wall();
    {
        la3dm_bgk_scan s = packed(0);
        la3dm_bgk_counters c;
        if (run_scan(&s, &c) != LA3DM_OK)
            throw std::runtime_error(std::string("BGKOctoMap::insert_pointcloud: ") + la3dm_last_error(ctx));
        stats.n_tiles += c.n_tiles;
    }
    stats.t_device = wall() - t1;
    commit();
    stats.t_total = wall() - t0;
}

void BGKOctoMap::insert_pointcloud_device(const float *d_xyz, size_t n, const point3f &origin, float ds_resolution, float free_res,
                                          float max_range) {
    bind();
    const double t0 = wall();
    if (dmap == nullptr) throw std::runtime_error("BGKOctoMap::insert_pointcloud_device: the map is not device resident");
    const float o[3] = {origin.x(), origin.y(), origin.z()};
    la3dm_devmap_stats ds;
    if (la3dm_devmap_insert_pointcloud_device(dmap, d_xyz, (uint32_t)n, o, ds_resolution, free_res, max_range, &ds) != LA3DM_OK)
        throw std::runtime_error(std::string("BGKOctoMap::insert_pointcloud_device: ") + la3dm_last_error(ctx));
    take_device_stats(ds);
    stats.t_total = wall() - t0;
    mirror_dirty = true;
}

void BGKOctoMap::insert_training_data(const GPPointCloud &cloud) {
    bind();
    const double t0 = wall();
    if (ctx == nullptr) throw std::runtime_error("BGKOctoMap::insert_training_data: no device context (there is no CPU path)");
    std::vector<float> flat;
    flat.reserve(cloud.size() * 4);
    for (const GPPointType &p : cloud) flat.insert(flat.end(), {p.first.x(), p.first.y(), p.first.z(), p.second});
    if (dmap != nullptr && variant != 3) {  // device-resident mode (a BGK-L map has no beams for a labelled set)
        la3dm_devmap_stats ds;
        if (la3dm_devmap_insert_training_data_host(dmap, flat.data(), (uint32_t)cloud.size(), &ds) != LA3DM_OK)
            throw std::runtime_error(std::string("BGKOctoMap::insert_training_data: ") + la3dm_last_error(ctx));
        take_device_stats(ds);
        stats.t_total = wall() - t0;
        mirror_dirty = true;
        return;
    }
    if (!prepare_training_data(flat.data(), cloud.size(), true)) return;
    const double t1 = wall();
    {
        la3dm_bgk_scan s = packed(0);
        if (run_scan(&s, nullptr) != LA3DM_OK)
            throw std::runtime_error(std::string("BGKOctoMap::insert_training_data: ") + la3dm_last_error(ctx));
    }
    stats.t_device = wall() - t1;
    commit();
    stats.t_total = wall() - t0;
}

}  // namespace la3dm

extern "C" void la3dm_node_ab(const void *node, float *A, float *B) {
    const la3dm::Occupancy *n = static_cast<const la3dm::Occupancy *>(node);
    *A = n->m_A;
    *B = n->m_B;
}
