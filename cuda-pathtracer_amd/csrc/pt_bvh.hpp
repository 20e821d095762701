// Host-side sphere BVH (see pt_bvh.cpp) and its device layout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace ptmi {

constexpr uint32_t kBvhLeafSize = 4;     // sphere slots per leaf
constexpr uint32_t kBvhMaxBig = 64;      // always-tested spheres kept beside the tree
constexpr uint32_t kBvhMaxDepth = 28;    // inner levels == entries of the per-lane traversal stack

struct BvhData {
    // Binary tree, four float4 per inner node holding the boxes of BOTH children:
    //   [0] = (lo0.x, lo0.y, lo0.z, hi0.x)   [1] = (hi0.y, hi0.z, lo1.x, lo1.y)
    //   [2] = (lo1.z, hi1.x, hi1.y, hi1.z)   [3] = bits (child0, child1, 0, 0)
    // A child reference >= 0 is an inner node index, < 0 is ~leaf.  Node 0 is the root.
    std::vector<float4> nodes;
    std::vector<float4> leaf_sph;  // [leaf][kBvhLeafSize] spheres (cx, cy, cz, r*r); unused slots can never be hit
    std::vector<int> leaf_index;   // [leaf][kBvhLeafSize] their original indices
    std::vector<float4> big_sph;   // always-tested spheres
    std::vector<int> big_index;
    uint32_t depth;                // inner levels on the longest root-to-leaf path
    float centre[3];               // centre of the bounded spheres' cloud
    float half_diag;               // every bounded sphere lies within this distance of `centre`
    float r_min;                   // smallest radius among the bounded spheres
};

// sph[i] = (cx, cy, cz, r*r), radius[i] = r.  Returns false when a hierarchy makes no sense
// (too few boundable spheres, or too many unboundable ones); the caller then scans brute force.
bool build_bvh(const float4 *sph, const float *radius, uint32_t n, BvhData *out);

} // namespace ptmi
