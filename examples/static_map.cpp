// static_map.cpp — the la3dm::BGKOctoMap class used from C++ the way the reference's static mapping node uses it
// (src/bgkoctomap/bgkoctomap_static_node.cpp:86-139: construct, insert N scans, get_bbox, walk the leaves), without
// ROS and PCL: the .pcd scans are read by the few lines of code below, the "publishing" is a count per state.
//
//   static_map <dir> <prefix> <scan_num> [resolution block_depth sf2 ell free_res ds_res max_range
//                                         free_thresh occupied_thresh var_thresh prior_A prior_B]
// prints:  leaves <n> occupied <n> free <n> unknown <n> blocks <n> bbox <min xyz> <max xyz> checksum <fnv-1a over (x, y, z, p, size) of the leaves in iteration order>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <algorithm>
#include <vector>

#include "../la3dm_amd/csrc/host/bgkoctomap.h"

// PCD v0.7, "DATA ascii" or "DATA binary" (what pcl::io::loadPCDFile is used for in the reference node): the
// fields x, y, z are located through FIELDS / SIZE / COUNT, VIEWPOINT tx ty tz ... gives the sensor origin.
static bool load_pcd(const std::string &path, la3dm::point3f &origin, la3dm::BGKOctoMap::PointCloud &cloud) {
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::vector<std::string> fields;
    std::vector<size_t> sizes, counts;
    size_t points = 0;
    std::string line, kind;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string key, tok;
        ss >> key;
        if (key == "FIELDS") while (ss >> tok) fields.push_back(tok);
        else if (key == "SIZE") while (ss >> tok) sizes.push_back((size_t)std::stoul(tok));
        else if (key == "COUNT") while (ss >> tok) counts.push_back((size_t)std::stoul(tok));
        else if (key == "POINTS") ss >> points;
        else if (key == "VIEWPOINT") {
            float x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            origin = la3dm::point3f(x, y, z);
        } else if (key == "DATA") {
            ss >> kind;
            break;
        }
    }
    if (fields.empty() || sizes.size() != fields.size()) return false;
    if (counts.size() != fields.size()) counts.assign(fields.size(), 1);
    size_t off[3] = {0, 0, 0}, idx[3] = {0, 0, 0}, stride = 0;
    bool have[3] = {false, false, false};
    for (size_t f = 0; f < fields.size(); ++f) {
        for (int a = 0; a < 3; ++a)
            if (fields[f] == std::string(1, "xyz"[a]) && sizes[f] == 4) {
                off[a] = stride;
                idx[a] = f;
                have[a] = true;
            }
        stride += sizes[f] * counts[f];
    }
    if (!have[0] || !have[1] || !have[2]) return false;
    cloud.clear();
    cloud.reserve(points);
    if (kind == "binary") {
        std::vector<char> rec(stride);
        for (size_t i = 0; i < points && in.read(rec.data(), (std::streamsize)stride); ++i) {
            float v[3];
            for (int a = 0; a < 3; ++a) std::memcpy(&v[a], rec.data() + off[a], 4);
            cloud.emplace_back(v[0], v[1], v[2]);
        }
    } else if (kind == "ascii") {
        while (cloud.size() < points && std::getline(in, line)) {
            std::istringstream ss(line);
            std::vector<float> row;
            float t;
            while (ss >> t) row.push_back(t);
            if (row.size() > std::max(idx[0], std::max(idx[1], idx[2]))) cloud.emplace_back(row[idx[0]], row[idx[1]], row[idx[2]]);
        }
    } else {
        return false;
    }
    return cloud.size() == points;
}

int main(int argc, char **argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: %s dir prefix scan_num [resolution block_depth sf2 ell free_res ds_res max_range ...]\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[1], prefix = argv[2];
    const int scan_num = std::atoi(argv[3]);
    float v[12] = {0.1f, 3, 1.0f, 0.2f, 0.5f, 0.1f, 8.0f, 0.3f, 0.7f, 100.0f, 0.001f, 0.001f};  // bgkoctomap.yaml + sim_structured.yaml
    for (int i = 0; i < 12 && 4 + i < argc; ++i) v[i] = (float)std::atof(argv[4 + i]);
    const float resolution = v[0], sf2 = v[2], ell = v[3], free_resolution = v[4], ds_resolution = v[5], max_range = v[6];
    const unsigned short block_depth = (unsigned short)v[1];
    try {
        la3dm::BGKOctoMap map(resolution, block_depth, sf2, ell, v[7], v[8], v[9], v[10], v[11]);
        for (int scan_id = 1; scan_id <= scan_num; ++scan_id) {
            la3dm::BGKOctoMap::PointCloud cloud;
            la3dm::point3f origin;
            const std::string filename = dir + "/" + prefix + "_" + std::to_string(scan_id) + ".pcd";
            if (!load_pcd(filename, origin, cloud)) {
                std::fprintf(stderr, "cannot read %s\n", filename.c_str());
                return 1;
            }
            map.insert_pointcloud(cloud, origin, ds_resolution, free_resolution, max_range);
        }
        la3dm::point3f lim_min, lim_max;
        map.get_bbox(lim_min, lim_max);
        uint64_t n = 0, occ = 0, fre = 0, unk = 0, h = 1469598103934665603ull;
        auto mix = [&](const void *p, size_t bytes) {
            const unsigned char *b = static_cast<const unsigned char *>(p);
            for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
        };
        for (auto it = map.begin_leaf(); it != map.end_leaf(); ++it) {
            ++n;
            const la3dm::State st = it.get_node().get_state();
            occ += st == la3dm::State::OCCUPIED;
            fre += st == la3dm::State::FREE;
            unk += st == la3dm::State::UNKNOWN;
            const la3dm::point3f p = it.get_loc();
            const float rec[5] = {p.x(), p.y(), p.z(), it.get_node().get_prob(), it.get_size()};
            mix(rec, sizeof(rec));
        }
        std::printf("leaves %llu occupied %llu free %llu unknown %llu blocks %zu bbox %g %g %g %g %g %g device_resident %d checksum %016llx\n",
                    (unsigned long long)n, (unsigned long long)occ, (unsigned long long)fre, (unsigned long long)unk,
                    map.block_count(), lim_min.x(), lim_min.y(), lim_min.z(), lim_max.x(), lim_max.y(), lim_max.z(),
                    map.is_device_resident() ? 1 : 0, (unsigned long long)h);
        // the node's publish step (bgkoctomap_static_node.cpp:101-136): cube lists per marker level, occupied cells
        // coloured by height, free cells by probability — scanned on the GPU when the map is device resident
        la3dm::BGKOctoMap::Cells occupied, free_cells;
        map.export_cells(la3dm::State::OCCUPIED, true, 0.0f, 0.0f, occupied);
        map.export_cells(la3dm::State::FREE, true, 0.0f, 0.0f, free_cells);
        for (const auto *c : {&occupied, &free_cells}) {
            size_t per_level[10] = {0};
            for (int32_t l : c->level) ++per_level[l < 10 ? l : 9];
            std::printf("%s cubes %zu by level:", c == &occupied ? "occupied" : "free", c->level.size());
            for (int l = 0; l < 10; ++l)
                if (per_level[l]) std::printf(" [%d] %zu", l, per_level[l]);
            std::printf("\n");
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
