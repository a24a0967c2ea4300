// ref_harness.cpp — C entry points over the REFERENCE's own std-only sources.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it includes the
// reference headers where they lie (/root/reference/include/...) and is linked with
// the reference's own point3f.cpp, bgkoctree_node.cpp, bgkoctree.cpp, bgkblock.cpp
// (see oracle/Makefile, target `ref`).  The result, oracle/_ref/libla3dm_ref.so,
// pins the oracle's restatement of block hashing, the voxel LUT, leaf order,
// Occupancy::update, OcTree::prune and the R-tree closed-box rule.
//
// Built a second time with -DLA3DM_REF_LV over the reference's BGK-LV family (bgklvoctree_node.cpp, bgklvoctree.cpp,
// bgklvblock.cpp, point6f.cpp — the same class names in the same namespace, hence a separate library,
// oracle/_ref/libla3dm_ref_lv.so): the LV node (min_W floor, f64 pow variance, UNCERTAIN state), the 28-bit node key,
// the LV block / tree and point6f's constructors.
//
// The reference's Eigen/PCL/ROS-dependent files (bgkinference.h, bgkoctomap.cpp, the
// nodes) are NOT buildable here (no Eigen/PCL/ROS in the image) and are not used.
//
// Access to the reference's private statics goes through the friendship the
// reference itself grants to `la3dm::BGKOctoMap` (bgkblock.h:56, bgkoctree.h:30,
// bgkoctree_node.h:28; the LV headers name `la3dm::BGKLVOctoMap`, bgklvoctree_node.h:29): this harness defines a class
// of that name.

#include <cmath>
#include <cstdint>
#include <vector>

#ifdef LA3DM_REF_LV
#include "bgklvblock.h"
#include "point6f.h"
#define REF_MAP_CLASS BGKLVOctoMap   // the friend the LV headers name (bgklvblock.h:56, bgklvoctree.h:30, bgklvoctree_node.h:29)
typedef unsigned long ref_index_t;   // bgklvoctree.h:15-18
#else
#include "bgkblock.h"
#define REF_MAP_CLASS BGKOctoMap
typedef unsigned short ref_index_t;
#endif
#include "rtree.h"

namespace la3dm {
class REF_MAP_CLASS {
public:
#ifdef LA3DM_REF_LV
    // the two extra statics of BGKLVOctoMap::BGKLVOctoMap, src/bgklvoctomap/bgklvoctomap.cpp:61-62
    static void configure_lv(bool original_size, float min_W) {
        OcTreeNode::original_size = original_size;
        OcTreeNode::min_W = min_W;
    }
#endif
    // what BGKOctoMap::BGKOctoMap does, src/bgkoctomap/bgkoctomap.cpp:31-56
    static void configure(float resolution, unsigned short block_depth, float sf2, float ell, float free_thresh,
                          float occupied_thresh, float var_thresh, float prior_A, float prior_B) {
        Block::resolution = resolution;
        Block::size = (float) pow(2, block_depth - 1) * resolution;
        Block::key_loc_map = init_key_loc_map(resolution, block_depth);
        Block::index_map = init_index_map(Block::key_loc_map, block_depth);
        OcTree::max_depth = block_depth;
        OcTreeNode::sf2 = sf2;
        OcTreeNode::ell = ell;
        OcTreeNode::free_thresh = free_thresh;
        OcTreeNode::occupied_thresh = occupied_thresh;
        OcTreeNode::var_thresh = var_thresh;
        OcTreeNode::prior_A = prior_A;
        OcTreeNode::prior_B = prior_B;
    }
    static float block_size() { return Block::size; }
    static int lut_size() { return (int) Block::key_loc_map.size(); }
    static bool lut(int key, float *out) {
        auto it = Block::key_loc_map.find(key);
        if (it == Block::key_loc_map.end()) return false;
        out[0] = it->second.x(); out[1] = it->second.y(); out[2] = it->second.z();
        return true;
    }
    static float A(const OcTreeNode &n) { return n.m_A; }
    static float B(const OcTreeNode &n) { return n.m_B; }
    static bool layer_alive(const OcTree &t, int depth) { return t.node_arr != nullptr && t.node_arr[depth] != nullptr; }
};
}  // namespace la3dm

using namespace la3dm;
typedef REF_MAP_CLASS BGKOctoMapShim;

extern "C" {

void ref_configure(float resolution, int block_depth, float sf2, float ell, float free_thresh, float occupied_thresh,
                   float var_thresh, float prior_A, float prior_B) {
    BGKOctoMapShim::configure(resolution, (unsigned short) block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh,
                          prior_A, prior_B);
}
float ref_block_size() { return BGKOctoMapShim::block_size(); }
int ref_sizeof_node() { return (int) sizeof(OcTreeNode); }
int ref_sizeof_block() { return (int) sizeof(Block); }
int ref_lut_size() { return BGKOctoMapShim::lut_size(); }
int ref_lut(int depth, int index, float *out3) { return BGKOctoMapShim::lut(node_to_hash_key(depth, index), out3); }

int64_t ref_block_to_hash_key(float x, float y, float z) { return block_to_hash_key(x, y, z); }
void ref_hash_key_to_block(int64_t key, float *out3) {
    point3f c = hash_key_to_block(key);
    out3[0] = c.x(); out3[1] = c.y(); out3[2] = c.z();
}
void ref_get_extended_block(int64_t key, int64_t *out7) {
    ExtendedBlock eb = get_extended_block(key);
    for (int i = 0; i < 7; ++i) out7[i] = eb[i];
}

void *ref_block_new(float cx, float cy, float cz) { return new Block(point3f(cx, cy, cz)); }
void ref_block_free(void *b) { delete (Block *) b; }
void ref_block_extended(void *b, int64_t *out7) {
    ExtendedBlock eb = ((Block *) b)->get_extended_block();
    for (int i = 0; i < 7; ++i) out7[i] = eb[i];
}
// grid view of a block (bgkblock.cpp:131-150): cell of p, its node key, its centre
void ref_block_grid(void *b, const float *p3, int32_t *idx3, int32_t *node_key, float *point3) {
    Block *blk = (Block *) b;
    unsigned short x, y, z;
    blk->get_index(point3f(p3[0], p3[1], p3[2]), x, y, z);
    idx3[0] = x; idx3[1] = y; idx3[2] = z;
    *node_key = blk->get_node(x, y, z);
    point3f q = blk->get_point(x, y, z);
    point3[0] = q.x(); point3[1] = q.y(); point3[2] = q.z();
}
int ref_block_leaves(void *b, int32_t *keys, float *loc_xyz, float *sizes, int cap) {
    Block *blk = (Block *) b;
    int n = 0;
    for (auto it = blk->begin_leaf(); it != blk->end_leaf(); ++it, ++n) {
        if (n < cap) {
            keys[n] = it.get_hash_key();
            point3f p = blk->get_loc(it);
            loc_xyz[3 * n] = p.x(); loc_xyz[3 * n + 1] = p.y(); loc_xyz[3 * n + 2] = p.z();
            sizes[n] = blk->get_size(it);
        }
    }
    return n;
}
void ref_block_update(void *b, int32_t key, float ybar, float kbar) { (*(Block *) b)[key].update(ybar, kbar); }
int ref_block_prune(void *b) { return ((Block *) b)->prune() ? 1 : 0; }
int ref_block_node(void *b, int32_t key, float *A, float *B, uint8_t *state, float *prob, float *var) {
    Block *blk = (Block *) b;
    unsigned short depth;
    ref_index_t index;
    hash_key_to_node(key, depth, index);
    if (!BGKOctoMapShim::layer_alive(*blk, depth)) return 0;
    OcTreeNode &n = (*blk)[key];
    *A = BGKOctoMapShim::A(n); *B = BGKOctoMapShim::B(n); *state = (uint8_t) n.get_state();
    *prob = n.get_prob(); *var = n.get_var();
    return 1;
}

// standalone node sequence: start from a default node, apply updates, report each step
void ref_node_sequence(const float *ybar, const float *kbar, int n, float *A, float *B, uint8_t *state, float *prob,
                       float *var) {
    OcTreeNode node;
    for (int i = 0; i < n; ++i) {
        node.update(ybar[i], kbar[i]);
        A[i] = BGKOctoMapShim::A(node); B[i] = BGKOctoMapShim::B(node); state[i] = (uint8_t) node.get_state();
        prob[i] = node.get_prob(); var[i] = node.get_var();
    }
}

#ifdef LA3DM_REF_LV
void ref_configure_lv(int original_size, float min_W) { BGKOctoMapShim::configure_lv(original_size != 0, min_W); }
// node_to_hash_key / hash_key_to_node with the 28-bit index, src/bgklvoctomap/bgklvoctree.cpp:9-16
int32_t ref_node_to_hash_key(int depth, uint32_t index) { return node_to_hash_key((unsigned short) depth, (unsigned long) index); }
void ref_hash_key_to_node(int32_t key, int32_t *depth, uint32_t *index) {
    unsigned short d;
    unsigned long i;
    hash_key_to_node(key, d, i);
    *depth = d; *index = (uint32_t) i;
}
// Occupancy(A, B): prior + (A, B), classified by the constructor (bgklvoctree_node.cpp:17-27); get_prob / get_var of it
void ref_node_ctor(float A, float B, float *mA, float *mB, uint8_t *state, float *prob, float *var) {
    OcTreeNode n(A, B);
    *mA = BGKOctoMapShim::A(n); *mB = BGKOctoMapShim::B(n); *state = (uint8_t) n.get_state();
    *prob = n.get_prob(); *var = n.get_var();
}
// point6f's constructors (include/common/point6f.h:43-92) as the LV front end uses them, and start() / end()
void ref_point6f(const float *a3, const float *b3, float *from_point, float *from_pair, float *from_xyz, float *start_end) {
    point3f a(a3[0], a3[1], a3[2]), b(b3[0], b3[1], b3[2]);
    point6f p(a), q(a, b), r(a3[0], a3[1], a3[2]);
    for (int i = 0; i < 6; ++i) { from_point[i] = p(i); from_pair[i] = q(i); from_xyz[i] = r(i); }
    point3f s = q.start(), e = q.end();
    start_end[0] = s.x(); start_end[1] = s.y(); start_end[2] = s.z();
    start_end[3] = e.x(); start_end[4] = e.y(); start_end[5] = e.z();
}
#endif

// R-tree with the reference's instantiation shape (bgkoctomap.h:32 uses GPPointType*; ids suffice)
typedef RTree<int, float, 3, float> RefRTree;
static bool collect_cb(int id, void *arg) {
    ((std::vector<int> *) arg)->push_back(id);
    return true;
}
void *ref_rtree_new(const float *xyz, int n) {
    RefRTree *t = new RefRTree;
    for (int i = 0; i < n; ++i) {
        float p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        t->Insert(p, p, i);
    }
    return t;
}
void ref_rtree_free(void *t) { delete (RefRTree *) t; }
// box query as BGKOctoMap::get_gp_points_in_bbox(key, out) does it
// (src/bgkoctomap/bgkoctomap.cpp:497-517): centre -/+ half_size in point3f arithmetic.
int ref_rtree_block_query(void *t, int64_t key, int32_t *ids, int cap) {
    float bs = BGKOctoMapShim::block_size();
    point3f half_size(bs / 2.0f, bs / 2.0f, bs / 2.0);
    point3f lim_min = hash_key_to_block(key) - half_size;
    point3f lim_max = hash_key_to_block(key) + half_size;
    float a_min[] = {lim_min.x(), lim_min.y(), lim_min.z()};
    float a_max[] = {lim_max.x(), lim_max.y(), lim_max.z()};
    std::vector<int> out;
    ((RefRTree *) t)->Search(a_min, a_max, collect_cb, &out);
    int n = (int) out.size();
    for (int i = 0; i < n && i < cap; ++i) ids[i] = out[i];
    return n;
}
int ref_rtree_box_query(void *t, const float *lo, const float *hi, int32_t *ids, int cap) {
    float a_min[] = {lo[0], lo[1], lo[2]};
    float a_max[] = {hi[0], hi[1], hi[2]};
    std::vector<int> out;
    ((RefRTree *) t)->Search(a_min, a_max, collect_cb, &out);
    int n = (int) out.size();
    for (int i = 0; i < n && i < cap; ++i) ids[i] = out[i];
    return n;
}

}  // extern "C"
