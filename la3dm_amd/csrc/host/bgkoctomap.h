// bgkoctomap.h — host side of the MI355X BGKOctoMap.
//
// Same public surface and data layout as the reference's BGKOctoMap family
// (include/bgkoctomap/{bgkoctomap,bgkblock,bgkoctree,bgkoctree_node}.h,
// include/common/point3f.h) so that the ROS nodes' call sequence
// (src/bgkoctomap/bgkoctomap_static_node.cpp:86-139) works unchanged:
//
//     la3dm::BGKOctoMap map(resolution, block_depth, sf2, ell, free_thresh,
//                           occupied_thresh, var_thresh, prior_A, prior_B);
//     map.insert_pointcloud(cloud, origin, ds_resolution, free_res, max_range);
//     for (auto it = map.begin_leaf(); it != map.end_leaf(); ++it) { it.get_loc(); it.get_node()...}
//
// What is different underneath: block hashing, neighbour gather (a counting sort by
// block instead of an R-tree) and octree bookkeeping stay on the host in this file's
// implementation; kernel evaluation and the alpha/beta fusion run on the GPU through
// the C ABI of include/la3dm_hip.h.  There is no CPU inference path: constructing a
// map without a HIP device throws std::runtime_error.
#ifndef LA3DM_AMD_BGKOCTOMAP_H
#define LA3DM_AMD_BGKOCTOMAP_H

#include <array>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <fstream>
#include <ostream>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../../include/la3dm_hip.h"

extern "C" void la3dm_node_ab(const void *node, float *A, float *B);  // (m_A, m_B) of an Occupancy

namespace la3dm {

/// 3 floats; the scalar type of every coordinate (reference: include/common/point3f.h).
class point3f {
public:
    point3f() : d{0.f, 0.f, 0.f} {}
    point3f(float x, float y, float z) : d{x, y, z} {}
    float &x() { return d[0]; }
    float &y() { return d[1]; }
    float &z() { return d[2]; }
    const float &x() const { return d[0]; }
    const float &y() const { return d[1]; }
    const float &z() const { return d[2]; }
    float &operator()(unsigned i) { return d[i]; }
    const float &operator()(unsigned i) const { return d[i]; }
    point3f operator+(const point3f &o) const { return point3f(d[0] + o.d[0], d[1] + o.d[1], d[2] + o.d[2]); }
    point3f operator-(const point3f &o) const { return point3f(d[0] - o.d[0], d[1] - o.d[1], d[2] - o.d[2]); }
    point3f operator*(float s) const { return point3f(d[0] * s, d[1] * s, d[2] * s); }
    void operator+=(const point3f &o) { d[0] += o.d[0]; d[1] += o.d[1]; d[2] += o.d[2]; }
    void operator-=(const point3f &o) { d[0] -= o.d[0]; d[1] -= o.d[1]; d[2] -= o.d[2]; }
    bool operator==(const point3f &o) const { return d[0] == o.d[0] && d[1] == o.d[1] && d[2] == o.d[2]; }
    /// double sqrt of a float sum, as the reference (point3f.h:207-214)
    double norm() const { return std::sqrt((double)(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])); }

private:
    float d[3];
};

typedef int OcTreeHashKey;     // (depth << 16) + index
typedef int64_t BlockHashKey;  // 3 x 20-bit block indices
typedef std::array<BlockHashKey, 7> ExtendedBlock;  // self,+x,-x,+y,-y,+z,-z

/// UNCERTAIN is used by the BGKLV variant only (reference bgklvoctree_node.h:11-13 numbers it 3 and PRUNED 4; the
/// C view and the device use those codes for LV maps, this enum is symbolic).
enum class State : char { FREE, OCCUPIED, UNKNOWN, PRUNED, UNCERTAIN };

inline OcTreeHashKey node_to_hash_key(unsigned short depth, unsigned short index) { return (depth << 16) + index; }
inline void hash_key_to_node(OcTreeHashKey key, unsigned short &depth, unsigned short &index) {
    depth = (unsigned short)(key >> 16);
    index = (unsigned short)(key & 0xFFFF);
}

/// Beta posterior of one voxel. 16 bytes: classified@0, m_A@4, m_B@8, state@12
/// (reference: include/bgkoctomap/bgkoctree_node.h:76-81).
class Occupancy {
    friend class BGKOctoMap;
    friend class BGKLVOctoMap;
    friend class OcTree;
    friend void ::la3dm_node_ab(const void *node, float *A, float *B);

public:
    /// BGK: (prior_A, prior_B); GP: (m_ivar, ivar) = (0, min_ivar)  (gpoctree_node.h:34)
    Occupancy() : classified(false), m_A(init_A), m_B(init_B), state(State::UNKNOWN) {}
    Occupancy(float A, float B);
    // like the reference, copying does not carry `classified`
    Occupancy(const Occupancy &o) : m_A(o.m_A), m_B(o.m_B), state(o.state) {}
    Occupancy &operator=(const Occupancy &o) {
        m_A = o.m_A;
        m_B = o.m_B;
        state = o.state;
        return *this;
    }

    /// Host twin of the device-side update (kept for API parity; the map itself
    /// never calls it — the GPU does the updates).
    void update(float ybar, float kbar);

    float get_prob() const;
    float get_var() const;
    State get_state() const { return state; }
    void prune() { state = State::PRUNED; }
    bool operator==(const Occupancy &rhs) const {
        return state != State::UNKNOWN && state != State::UNCERTAIN && state == rhs.state;
    }

    /// The reference node's stream operators (src/bgkoctomap/bgkoctree_node.cpp:46-62; unused by its own nodes): the binary
    /// pair writes m_A, m_B as 8 raw bytes and reads them back THROUGH the (A, B) constructor — which adds the priors again,
    /// as the reference does —; the text form is "(m_A m_B prob)".
    friend std::ofstream &operator<<(std::ofstream &os, const Occupancy &oc);
    friend std::ifstream &operator>>(std::ifstream &is, Occupancy &oc);
    friend std::ostream &operator<<(std::ostream &os, const Occupancy &oc);

    bool classified;

private:
    void classify();
    float m_A;
    float m_B;
    State state;

    static float sf2, ell, prior_A, prior_B, free_thresh, occupied_thresh, var_thresh;
    // GPOctoMap statics (src/gpoctomap/gpoctree_node.cpp:7-17); for GP nodes m_A holds m_ivar, m_B holds ivar
    static int variant;  // 0 BGK, 1 GP, 2 BGKLV
    static float min_W;  // BGKLV (bgklvoctree_node.cpp:15)
    static bool original_size;
    static float init_A, init_B, noise, l, max_ivar, min_ivar, min_known_ivar;
};
typedef Occupancy OcTreeNode;
static_assert(sizeof(Occupancy) == 16, "node layout must match the reference (16 bytes)");

/// Fixed-depth test-data octree of one block: layer d holds 8^d nodes, child c of node i
/// is node 8i+c of the next layer (c&4 -> +x, c&2 -> +y, c&1 -> +z).
class OcTree {
    friend class BGKOctoMap;
    friend class BGKLVOctoMap;

public:
    OcTree();
    ~OcTree();
    OcTree(const OcTree &) = delete;
    OcTree &operator=(const OcTree &) = delete;

    bool prune();
    /// overwrite every node from flat (alpha, beta, state|classified<<7) arrays in depth-major node order
    void load_nodes(const float *A, const float *B, const uint8_t *S, size_t n);
    bool is_leaf(OcTreeHashKey key) const;
    bool is_leaf(unsigned short depth, unsigned short index) const;
    bool search(OcTreeHashKey key) const;
    OcTreeNode &operator[](OcTreeHashKey key) const;

    /// Leaves in the reference's LeafIterator order (depth-first, children 7..0).
    class LeafIterator {
    public:
        LeafIterator() : tree(nullptr), top(0) {}
        explicit LeafIterator(const OcTree *t);
        bool operator==(const LeafIterator &o) const {
            return tree == o.tree && top == o.top && (top == 0 || stack[top - 1] == o.stack[o.top - 1]);
        }
        bool operator!=(const LeafIterator &o) const { return !(*this == o); }
        LeafIterator &operator++();
        LeafIterator operator++(int) {
            LeafIterator r(*this);
            ++(*this);
            return r;
        }
        OcTreeNode &operator*() const { return (*tree)[get_hash_key()]; }
        OcTreeNode &get_node() const { return operator*(); }
        OcTreeHashKey get_hash_key() const { return stack[top - 1]; }

    private:
        void settle();
        const OcTree *tree;
        int top;
        OcTreeHashKey stack[7 * 6 + 2];  // depth <= 6: at most 7 siblings pending per level
    };
    LeafIterator begin_leaf() const { return LeafIterator(this); }
    LeafIterator end_leaf() const { return LeafIterator(); }

    /// Append the leaf keys in LeafIterator order (bulk form used by the packer).
    void collect_leaves(std::vector<uint32_t> &keys) const;

protected:
    OcTreeNode **node_arr;  // node_arr[d] == nullptr once a layer is fully collapsed
    OcTreeNode *slab;       // all layers in one allocation
    bool ever_pruned;
    static unsigned short max_depth;
};

BlockHashKey block_to_hash_key(point3f center);
BlockHashKey block_to_hash_key(float x, float y, float z);
point3f hash_key_to_block(BlockHashKey key);
ExtendedBlock get_extended_block(BlockHashKey key);

/// Voxel LUT, flat and depth-major: entry (d, i) at (8^d - 1)/7 + i
/// (replaces the reference's unordered_map Block::key_loc_map; same values).
std::vector<point3f> init_key_loc_map(float resolution, unsigned short max_depth);

class Block : public OcTree {
    friend class BGKOctoMap;
    friend class BGKLVOctoMap;
    friend BlockHashKey block_to_hash_key(float x, float y, float z);
    friend point3f hash_key_to_block(BlockHashKey key);
    friend ExtendedBlock get_extended_block(BlockHashKey key);

public:
    Block() : OcTree(), center(0.f, 0.f, 0.f) {}
    explicit Block(point3f c) : OcTree(), center(c) {}

    static const point3f &lut(OcTreeHashKey key) {
        return key_loc_map[(0x249249u & ((1u << (3u * (unsigned)(key >> 16))) - 1u)) + (unsigned)(key & 0xFFFF)];
    }
    point3f get_loc(const LeafIterator &it) const { return lut(it.get_hash_key()) + center; }
    float get_size(const LeafIterator &it) const { return float(size / pow(2, it.get_hash_key() >> 16)); }
    point3f get_center() const { return center; }
    point3f get_lim_min() const { return center - point3f(size / 2.0f, size / 2.0f, size / 2.0f); }
    point3f get_lim_max() const { return center + point3f(size / 2.0f, size / 2.0f, size / 2.0f); }
    ExtendedBlock get_extended_block() const;
    /// voxel containing p (index clamped into the block), at the finest layer
    OcTreeNode &search(point3f p) const;
    // ---- finest-layer grid view of a block (reference bgkblock.cpp:131-147, index_map :34-67) ----
    /// voxels per block edge.  (The reference keeps a static cell_num that is initialised once from the default
    /// statics — 8 — and never updated by the map constructor, so its grid view is only meaningful at
    /// block_depth 4; this is the value it has there.)
    static unsigned short cell_num() { return (unsigned short)(1u << (max_depth - 1)); }
    /// grid cell of p, clamped into the block; truncation toward zero as in the reference
    void get_index(const point3f &p, unsigned short &x, unsigned short &y, unsigned short &z) const;
    /// finest-layer node of grid cell (x, y, z): per level the child bit 4 is +x, 2 is +y, 1 is +z
    static OcTreeHashKey get_node(unsigned short x, unsigned short y, unsigned short z);
    point3f get_point(unsigned short x, unsigned short y, unsigned short z) const { return lut(get_node(x, y, z)) + center; }
    OcTreeNode &search(float x, float y, float z) const { return search(point3f(x, y, z)); }

private:
    static std::vector<point3f> key_loc_map;
    static float resolution;
    static float size;
    point3f center;
};

struct ScanStats {
    uint64_t n_hits = 0, n_frees = 0, n_bbox_blocks = 0, n_train_blocks = 0, n_test_blocks = 0;
    uint64_t voxel_updates = 0;  // U
    uint64_t train_reads = 0;    // sum_t sum_{b in E(t)} N_b
    uint64_t pair_evals = 0;     // P
    uint64_t n_tiles = 0;
    double t_frontend = 0, t_partition = 0, t_pack = 0, t_device = 0, t_commit = 0, t_prune = 0, t_total = 0;
    double t_gather = 0;   // sharded device-resident insert with LA3DM_TIMING=1: the all-gather-v of the leaves
};

class BGKOctoMap {
public:
    typedef std::vector<point3f> PointCloud;
    typedef std::pair<point3f, float> GPPointType;
    typedef std::vector<GPPointType> GPPointCloud;

    BGKOctoMap();
    virtual ~BGKOctoMap();
    /// Same argument order as the reference constructor (bgkoctomap.h:50-58); `device`
    /// is the HIP device ordinal (a negative ordinal builds a bookkeeping-only map without a
    /// GPU context — prepare()/packed()/commit() work, insert_* throw; used to test host logic).
    BGKOctoMap(float resolution, unsigned short block_depth, float sf2, float ell, float free_thresh,
               float occupied_thresh, float var_thresh, float prior_A, float prior_B, int device = 0);
    BGKOctoMap(const BGKOctoMap &) = delete;
    BGKOctoMap &operator=(const BGKOctoMap &) = delete;

protected:
    struct GPParams {  // variant-specific constructor arguments (GP: first five; BGKLV: last two)
        float noise, l, min_var, max_var, max_known_var;
        float min_W;
        bool original_size;
    };
    /// shared constructor: variant 0 = BGK (gp == nullptr), 1 = GP
    BGKOctoMap(int variant, float resolution, unsigned short block_depth, float sf2, float ell, float free_thresh,
               float occupied_thresh, float var_thresh, float prior_A, float prior_B, const GPParams *gp, int device);

public:
    float get_resolution() const { return resolution; }
    float get_block_depth() const { return block_depth; }
    float get_block_size() const { return block_size; }
    /// reference bgkoctomap.cpp:66-80 (the same pair in gpoctomap.cpp:55-69, bgkloctomap.cpp:67-81, bgklvoctomap.cpp:73-87):
    /// re-derive the block size and the voxel LUT for a new resolution / block depth.  The reference does this under
    /// whatever blocks the map already holds (their octrees keep the old depth, their centres the old block size — a
    /// filled map is silently corrupted); here it is legal on an EMPTY map only and throws std::logic_error otherwise.
    /// The device context (LUT, kernels' block depth) and the device-resident pool are rebuilt; options set through
    /// la3dm_set_option return to their defaults.
    void set_resolution(float resolution);
    void set_block_depth(unsigned short max_depth);

    /// One scan. xyz: n points, `stride` floats between consecutive points (3 for packed
    /// xyz, 4 for PCL's PointXYZ).  Mirrors insert_pointcloud(const PCLPointCloud&, ...)
    /// (bgkoctomap.h:82-84): ds_resolution < 0 disables the voxel-grid filter,
    /// max_range <= 0 disables the range gate. Empty training set => silent return.
    virtual void insert_pointcloud(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                                   float free_res = 2.0f, float max_range = -1);
    void insert_pointcloud(const PointCloud &cloud, const point3f &origin, float ds_resolution, float free_res = 2.0f,
                           float max_range = -1) {
        insert_pointcloud(cloud.empty() ? nullptr : &cloud[0].x(), cloud.size(), 3, origin, ds_resolution, free_res,
                          max_range);
    }
#if defined(LA3DM_WITH_PCL)
    template <class PCLCloud>
    void insert_pointcloud(const PCLCloud &cloud, const point3f &origin, float ds_resolution, float free_res = 2.0f,
                           float max_range = -1) {
        insert_pointcloud(cloud.empty() ? nullptr : &cloud.points[0].x, cloud.size(),
                          sizeof(cloud.points[0]) / sizeof(float), origin, ds_resolution, free_res, max_range);
    }
#endif
    /// Pre-labelled training points (y = 1 hit, 0 free); updates are not gated on kbar
    /// (reference bgkoctomap.cpp:82-212, without its null dereference).
    void insert_training_data(const GPPointCloud &xy);
    /// insert_pointcloud for a cloud that already lives in HBM (n packed xyz triples on the map's device): the
    /// device-resident mode only; nothing crosses PCIe but the scalars that size the launches.
    void insert_pointcloud_device(const float *d_xyz, size_t n, const point3f &origin, float ds_resolution,
                                  float free_res = 2.0f, float max_range = -1);

    void get_bbox(point3f &lim_min, point3f &lim_max) const;

    // ---- split form of insert_pointcloud, used by the benchmark and the multi-GPU
    // driver: prepare() runs the front end, the block partition and packs the scan into
    // flat host arrays (valid until the next prepare); the caller runs the kernel
    // (la3dm_bgk_scan_host / _device on any shard of the test blocks) and hands the
    // updated leaf arrays back to commit(), which writes the nodes and prunes.
    /// returns false when there is nothing to do (empty training set)
    bool prepare(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution, float free_res,
                 float max_range);
    bool prepare_training_data(const float *xyzy, size_t n, bool ungated);
    /// the packed scan (host pointers into this map's buffers): pass 0 = every distinct
    /// test block. (Repeated keys of get_blocks_in_bbox — normally none — form further
    /// passes that commit() replays itself.)
    size_t num_passes() const { return passes.size(); }
    la3dm_bgk_scan packed(size_t pass = 0);
    /// write the updated leaves of pass 0 back into the nodes, replay passes >= 1, prune
    void commit();
    la3dm_ctx *device_ctx() const { return ctx; }
    const ScanStats &last_stats() const { return stats; }

    // ---- device-resident mode: the block pool lives in HBM and a scan runs start
    // to finish on the GPU (front end, partition, predict + fuse, write-back, prune — include/la3dm_hip.h,
    // la3dm_devmap_*).  The host blocks become a mirror that is refreshed lazily, the first time a query
    // (search, begin_leaf, block_count) follows a scan; get_bbox, search_many and export_cells are answered from
    // the pool without a refresh.  This is the DEFAULT for BGK, GP and BGK-L maps with a GPU context (block_depth <= 5;
    // LA3DM_DEVICE_RESIDENT=0 disables it); insert_pointcloud and insert_training_data both run on the pool.  The
    // split prepare()/commit() form is host-orchestrated: calling it moves the map to the host mode for good (the
    // pool is downloaded once).  Results are bit-identical in both modes.
    void set_device_resident(bool on);
    void ensure_host_mode();
    bool is_device_resident() const { return dmap != nullptr; }
    /// Block-sharded insert_pointcloud across `world` GPUs (one process per GPU, every process holds a replica of the map
    /// and inserts the same clouds): see la3dm_devmap_set_shard in include/la3dm_hip.h.  Needs the device-resident mode.
    void set_shard(uint32_t rank, uint32_t world, la3dm_allgatherv_fn fn, void *user);
    void sync_mirror() const;
    void take_device_stats(const la3dm_devmap_stats &ds);
    /// training set (x, y, z, label) the device front end produced for the last scan
    std::vector<float> device_training_data() const;
    const std::vector<float> &last_training_data() const { return xy; }  // x,y,z,label

    /// Voxel walk along the segment start -> end at the base resolution (reference bgkoctomap.h:91-214):
    /// an integer DDA on the voxel indices of the two end points with one error term per axis pair, crossing
    /// into neighbour blocks (which may be missing: next() then reports valid = false).  Starts only if the
    /// block that holds `start` exists.
    class RayCaster {
    public:
        RayCaster(const BGKOctoMap *map, const point3f &start, const point3f &end);
        bool end() const { return n <= 0; }
        /// current voxel: centre (or the dead-reckoned position when its block is missing), a copy of its
        /// node, block and node key; then advance.  Returns whether the block exists.
        bool next(point3f &p, OcTreeNode &node, BlockHashKey &block_key, OcTreeHashKey &node_key);

    private:
        void enter_block(int axis, int inc);
        const BGKOctoMap *map;
        Block *block;
        point3f block_center, current_p;
        int idx[3], inc[3], d2[3];   // voxel index inside the block, step sign, 2 * |delta| per axis
        int err_xy, err_xz, err_yz, n, lim;
        BlockHashKey key;
    };

    class LeafIterator {
    public:
        explicit LeafIterator(const BGKOctoMap *map);
        LeafIterator(std::unordered_map<BlockHashKey, Block *>::const_iterator bit, OcTree::LeafIterator lit)
            : block_it(bit), end_block(bit), leaf_it(lit), end_leaf(lit) {}
        bool operator==(const LeafIterator &o) const { return block_it == o.block_it && leaf_it == o.leaf_it; }
        bool operator!=(const LeafIterator &o) const { return !(*this == o); }
        LeafIterator &operator++();
        OcTreeNode &operator*() const { return *leaf_it; }
        OcTreeNode &get_node() const { return *leaf_it; }
        point3f get_loc() const { return block_it->second->get_loc(leaf_it); }
        float get_size() const { return block_it->second->get_size(leaf_it); }
        BlockHashKey get_block_key() const { return block_it->first; }
        OcTreeHashKey get_node_key() const { return leaf_it.get_hash_key(); }
        std::vector<point3f> get_pruned_locs() const;

    private:
        std::unordered_map<BlockHashKey, Block *>::const_iterator block_it, end_block;
        OcTree::LeafIterator leaf_it, end_leaf;
    };
    LeafIterator begin_leaf() const {
        bind();
        sync_mirror();
        return LeafIterator(this);
    }
    LeafIterator end_leaf() const { return LeafIterator(block_arr.cend(), OcTree::LeafIterator()); }

    OcTreeNode search(point3f p) const;
    OcTreeNode search(float x, float y, float z) const { return search(point3f(x, y, z)); }
    Block *search(BlockHashKey key) const;
    /// search(x, y, z) for n points at once (packed xyz).  In device-resident mode the answers come straight from the
    /// device pool (no mirror refresh); otherwise from the host blocks.  exists[i] = the block exists.
    void search_many(const float *xyz, size_t n, uint8_t *exists, float *A, float *B, uint8_t *state) const;
    size_t block_count() const {
        bind();
        sync_mirror();
        return block_arr.size();
    }
    int get_variant() const { return variant; }

    /// Cube lists of the map = the publish loop of the static node (src/bgkoctomap/bgkoctomap_static_node.cpp:101-136)
    /// with MarkerArrayPub::insert_point3d / heightMapColor (include/common/markerarray_pub.h:21-147) minus ROS.
    /// `state` OCCUPIED: cells coloured by height between min_z and max_z (min_z == max_z: the map's bbox, as the node
    /// does); FREE: coloured by probability.  original_size false expands collapsed leaves (get_pruned_locs).
    /// level[i] = (int) log2(size / resolution) = the CUBE_LIST marker the cell goes to.  In device-resident mode the
    /// pool is scanned on the GPU (no mirror refresh).  Returns the number of cells.
    struct Cells {
        std::vector<float> xyz_size;  // 4 per cell
        std::vector<float> rgba;      // 4 per cell
        std::vector<int32_t> level;
    };
    size_t export_cells(State state, bool original_size, float min_z, float max_z, Cells &out) const;

protected:
    void get_training_data(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                           float free_resolution, float max_range);
    bool partition_and_pack(bool ungated);
    void refresh_pass(size_t p);
    void write_nodes(size_t p);

    // The reference keeps its map parameters in process-global statics (Block::resolution / size / key_loc_map,
    // OcTree::max_depth, every OcTreeNode threshold: bgkoctomap.cpp:31-56) and so do the node / block classes here.
    // Each map therefore remembers its own set and re-installs it (bind()) at the top of every public entry point, so
    // several maps with different variants, depths or resolutions can be alive in one process.  Like the reference the
    // maps of a process must be driven from one thread at a time, and a LeafIterator / RayCaster must be used up
    // before another map is touched.
    struct BoundStatics {
        float resolution, size;
        std::vector<point3f> key_loc_map;
        unsigned short max_depth;
        float sf2, ell, free_thresh, occupied_thresh, var_thresh, prior_A, prior_B, init_A, init_B, min_W;
        float noise, l, min_ivar, max_ivar, min_known_ivar;
        int variant;
        bool original_size;
    };
    BoundStatics mine;
    static const BGKOctoMap *bound;
    void capture_statics();

public:
    /// make this map's parameters the process-global ones (every public entry point does it; call it yourself before
    /// using the static helpers block_to_hash_key / hash_key_to_block / get_extended_block with a particular map in mind)
    void bind() const;

protected:

    float resolution;
    float block_size;
    unsigned short block_depth;
    mutable std::unordered_map<BlockHashKey, Block *> block_arr;
    la3dm_ctx *ctx;
    la3dm_devmap *dmap = nullptr;
    mutable bool mirror_dirty = false;
    la3dm_params create_params;   // what the context was created with (lut_xyz is re-pointed on use)
    void create_context();        // la3dm_create + the device-resident pool from create_params and the current statics
    void reconfigure(float resolution, unsigned short depth);
    struct ShardCfg {
        uint32_t rank = 0, world = 1;
        la3dm_allgatherv_fn fn = nullptr;
        void *user = nullptr;
    } shard_cfg;   // what set_shard() installed (re-applied to the device map reconfigure() builds)

    // per-scan buffers (capacity reused across scans)
    std::vector<float> xy;               // training set: x,y,z,label
    std::vector<float> train_xyzy;       // grouped by training block
    std::vector<uint32_t> train_off;
    struct Pass {
        std::vector<BlockHashKey> keys;
        std::vector<Block *> blocks;
        std::vector<int32_t> nbr;
        std::vector<float> center;
        std::vector<uint32_t> leaf_off, leaf_key;
        std::vector<float> alpha, beta;
        std::vector<uint8_t> state;
    };
    std::vector<Pass> passes;
    std::vector<BlockHashKey> prune_list;  // test_blocks in list order (with repeats)
    uint32_t scan_flags;
    ScanStats stats;
    int variant;
    uint32_t train_max_n;
    uint64_t train_sum_n2;
    int run_scan(la3dm_bgk_scan *s, la3dm_bgk_counters *c);

    // ---- variant 3 (BGKLOctoMap): training samples keep their beam, training blocks hold segment rows ----
    void get_training_data_l(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                             float free_resolution, float max_range);
    void build_rows_l();
    std::vector<int32_t> l_ray_idx;     // per training sample: -1 = hit, else index into l_rays
    std::vector<float> l_rays;          // 6 floats per beam: origin -> hit shortened by free_resolution
    std::vector<uint32_t> train_src;    // training sample index of every CSR member
    std::vector<float> train_rows;      // 8 floats per row {x0,y0,z0,x1,y1,z1,label,0}
    std::vector<uint32_t> rows_off;     // CSR over training blocks

public:
    const std::vector<int32_t> &last_ray_index() const { return l_ray_idx; }
    const std::vector<float> &last_rays() const { return l_rays; }
};

/// BGKLOctoMap: block-level BGK whose free-space evidence are the beams themselves (line segments) instead of
/// down-sampled free points (reference include/bgkloctomap/bgkloctomap.h, src/bgkloctomap/bgkloctomap.cpp:31-57).
/// Same node type and constructor argument order as BGKOctoMap.
class BGKLOctoMap : public BGKOctoMap {
public:
    BGKLOctoMap() : BGKLOctoMap(0.1f, 4, 1.0, 1.0, 0.3f, 0.7f, 1.0f, 1.0f, 1.0f) {}
    BGKLOctoMap(float resolution, unsigned short block_depth, float sf2, float ell, float free_thresh, float occupied_thresh,
                float var_thresh, float prior_A, float prior_B, int device = 0)
        : BGKOctoMap(3, resolution, block_depth, sf2, ell, free_thresh, occupied_thresh, var_thresh, prior_A, prior_B, nullptr,
                     device) {}
};

/// BGKLVOctoMap: variance-aware BGK with free-space line segments and per-voxel inference
/// (reference include/bgklvoctomap/bgklvoctomap.h; constructor order src/bgklvoctomap/bgklvoctomap.cpp:33-43).
/// Every block of the scan's bounding box is allocated, every base-resolution leaf is inferred on the GPU.
class BGKLVOctoMap : public BGKOctoMap {
public:
    BGKLVOctoMap() : BGKLVOctoMap(0.1f, 4, 1.0, 1.0, 0.3f, 0.7f, 1.0f, 1.0f, 1.0f, true, 0.1f) {}
    BGKLVOctoMap(float resolution, unsigned short block_depth, float sf2, float ell, float free_thresh, float occupied_thresh,
                 float var_thresh, float prior_A, float prior_B, bool original_size, float min_W, int device = 0);
    void insert_pointcloud(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                           float free_res = 2.0f, float max_range = -1) override;
    using BGKOctoMap::insert_pointcloud;

    struct LVStats {
        uint64_t n_hits = 0, n_rays = 0, n_samples = 0, n_bbox_blocks = 0, n_packed_blocks = 0, n_info_blocks = 0,
                 voxels = 0, voxel_updates = 0;
        double t_frontend = 0, t_partition = 0, t_device = 0, t_commit = 0, t_total = 0;
    };
    const LVStats &lv_stats() const { return lvst; }
    /// training samples (x, y, z, ray index or -1) and segments (6 floats) of the last scan
    const std::vector<float> &lv_samples() const {
        fetch_device_training();
        return samples;
    }
    const std::vector<float> &lv_rays() const {
        fetch_device_training();
        return rays6;
    }
    /// split form used by the benchmark: prepare_lv packs (returns false if nothing to do), packed_lv exposes the
    /// device-call arguments (host pointers), commit_lv writes the nodes and prunes (the split form runs the first visit
    /// of every block only; insert_pointcloud also runs the repeats of the float-stepped candidate list)
    bool prepare_lv(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution, float free_res,
                    float max_range);
    la3dm_lv_scan packed_lv();
    void commit_lv();
    void finish_lv();

private:
    void training_data_lv(const float *xyz, size_t n, size_t stride, const point3f &origin, float ds_resolution,
                          float free_resolution, float max_range);
    void fetch_device_training() const;   // device-resident mode: samples / segments of the last scan come from the GPU on demand
    mutable bool device_training_stale = false;
    mutable std::vector<float> samples;   // x, y, z, ray
    std::vector<float> sorted;    // x, y, z, original index bits
    std::vector<float> rays8;
    mutable std::vector<float> rays6;
    std::vector<uint32_t> cell_off;
    int32_t cell_min[3], cell_dim[3];
    std::vector<float> lv_center, lv_alpha, lv_beta;
    std::vector<int32_t> lv_cell0;
    std::vector<uint8_t> lv_state;
    std::vector<Block *> lv_blocks;
    // A block index the float-stepped candidate loop produces k times is visited k times by the reference (the same
    // samples, serially): lv_mult[b] = k; pass p re-packs and re-runs the blocks with k > p (lv_all_* keep the full
    // list, lv_blocks / lv_center / lv_cell0 hold the current pass), the prune follows the last pass.
    std::vector<Block *> lv_all_blocks;
    std::vector<float> lv_all_center;
    std::vector<int32_t> lv_all_cell0;
    std::vector<uint32_t> lv_mult;
    std::vector<uint8_t> lv_info;   // per entry of lv_all_blocks: some voxel had samples in its box (any pass)
    std::vector<uint32_t> lv_pass_index;  // lv_blocks[i] == lv_all_blocks[lv_pass_index[i]]
    uint32_t lv_max_mult = 0;
    bool select_pass_lv(uint32_t pass);
    LVStats lvst;
};

/// GPOctoMap: same skeleton, GP regression per block + BCM fusion (reference include/gpoctomap/gpoctomap.h).
/// Constructor argument order as the reference (src/gpoctomap/gpoctomap.cpp:23-25).
class GPOctoMap : public BGKOctoMap {
public:
    GPOctoMap() : GPOctoMap(0.1f, 4, 1.0, 1.0, 0.01, 100, 0.001f, 1000.0f, 0.02f, 0.3f, 0.7f) {}
    GPOctoMap(float resolution, unsigned short block_depth, float sf2, float ell, float noise, float l, float min_var,
              float max_var, float max_known_var, float free_thresh, float occupied_thresh, int device = 0);
};

}  // namespace la3dm
#endif
