// tn_build.h -- the device-side structure build of load_tetrahedra (tn_build.hip).
#pragma once
#include "tn_build_core.h"
#include "tn_devbuf.h"

namespace tn {

// shape of the median-split binary tree over n faces (tn_mesh.cpp)
void build_bin_topology(size_t n, std::vector<core::BinNode> &bn, std::vector<std::vector<uint32_t>> &frontier,
                        std::vector<uint32_t> &level_start, std::vector<uint32_t> &leaf_nodes, uint32_t leaf_w = WIDE);

struct BuildTargets {
    DevBuf<uint32_t> &faces, &face_tets;   // [F,3], [F,2]
    DevBuf<WalkVar> &vars;                 // [4T]
    DevBuf<float> &hull_nodes, &hull_tris;
    DevWideBvh &bvh;
};
struct BuildInfo {
    uint32_t F = 0, n_hull = 0, n_hull_nodes = 0, max_stack = 1;
    float scene_max = 0.f;
};
// Everything load_tetrahedra owns, built on the device from the caller's (device) xyz / cells on `stream`:
// face table in first-seen order, face -> tets, Morton-ordered walk records, hull tree, 64-wide face BVH.
// Blocking (a handful of small D2H reads: counts, the hull faces, the child rows for the stack bound).
// Throws the reference's errors ("A triangle is shared by more than two tetrahedra!", out-of-bounds vertex ids).
void device_build(size_t V, size_t T, const float *xyz, const uint32_t *cells, hipStream_t stream, BuildTargets out,
                  BuildInfo &info, uint32_t leaf_w = WIDE);

}  // namespace tn
