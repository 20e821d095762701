// Host-side sphere BVH (see pt_bvh.cpp) and its device layout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace ptmi {

constexpr uint32_t kBvhLeafSize = 4;     // sphere slots per leaf
constexpr uint32_t kBvhMaxBig = 64;      // always-tested spheres kept beside the tree
constexpr uint32_t kBvhMaxDepth = 28;    // inner levels == entries of the per-lane traversal stack

constexpr uint32_t kBvhWidth = 4;        // children per inner node
constexpr int kBvhNoChild = 0x7fffffff;  // reference of an unused child slot
constexpr uint32_t kBvhNodeVec4 = 8;     // float4 per inner node (128 bytes: one cache line)

struct BvhData {
    // Four-wide tree.  An inner node is eight float4 (one 128-byte line), the boxes of its up to
    // four children in structure-of-arrays order:
    //   [0] lo.x of children 0..3   [1] lo.y   [2] lo.z   [3] hi.x   [4] hi.y   [5] hi.z
    //   [6] bits: the four child references   [7] unused
    // A child reference >= 0 is an inner node index, < 0 is ~leaf, kBvhNoChild marks an unused slot
    // (its box is all NaN).  Node 0 is the root.  The tree is the binary surface-area tree with
    // every other level folded into its parent: half as many dependent node fetches per ray.
    std::vector<float4> nodes;
    // The same nodes as the traversal reads them: four uint4 (64 bytes) per node, every box plane a
    // 16-bit coordinate on a grid over the bounded spheres' extent (plane = grid_origin +
    // q * grid_step per axis), lower planes rounded down and upper planes up, two more steps
    // outwards to cover the rounding of the traversal's plane arithmetic:
    //   [0] lo.x of children 0..3 (two per word, child 0 in the low half), lo.y likewise
    //   [1] lo.z, hi.x   [2] hi.y, hi.z   [3] the four child references
    // Halves the bytes and nearly halves the load instructions per node visit, which is what the
    // traversal is bound by (measured: three extra loads per visit cost 19 %).
    std::vector<uint4> qnodes;
    float grid_origin[3], grid_step[3];
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
