// Host-side build of the sphere BVH used by the accelerated nearest-hit search.
//
// The reference only has a brute-force scan ("TODO kd-tree", scene.h:33).  The hierarchy must
// return exactly what that scan returns -- the lexicographic minimum of (t, index) over the
// spheres whose fp32 test (primitive.h:39-45) yields a valid t -- so it may only skip spheres
// that provably cannot produce a valid t below the current bound.  The fp32 test can report hits
// off the geometric sphere: with op = centre - origin and a direction of squared length 1 + eps
// (mirror directions are not re-normalised, globals.h:107-109), the point origin + t * dir of a
// reported t lies at squared distance r^2 + eps * t^2 + err from the centre, where err, the
// rounding error of det = b*b - op.op + r*r, stays below 1e-6 * |op|^2 * |dir|^2.  Both terms depend on
// the ray, so the traversal widens every box by a per-ray margin derived from them (pt_trace.hpp,
// intersect_bvh); the boxes built here bound the exact spheres plus a small fixed slack.
// Spheres too large to bound usefully (the demo room's radius-1e5 walls, the light) are kept in
// a separate always-tested list.
#include "pt_bvh.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace ptmi {

namespace {

struct Item {
    float c[3];
    float R;     // radius plus slack
    int index;   // original sphere index
};

// A subtree of the binary surface-area tree, before every other level is folded away.
struct Subtree {
    int ref;          // >= 0 index into Builder::binary, < 0 ~leaf
    float lo[3], hi[3];
};

struct BinaryNode {
    Subtree child[2];
};

struct Builder {
    std::vector<Item> items;
    std::vector<BinaryNode> binary;
    const float4 *sph;
    BvhData *out;

    static float half_area(const float lo[3], const float hi[3])
    {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }

    void sort_axis(size_t lo, size_t hi, int axis)
    {
        std::sort(items.begin() + lo, items.begin() + hi, [axis](const Item &x, const Item &y) {
            return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.index < y.index);
        });
    }

    // Where to cut items[lo, hi) (sorted along the returned axis on return): surface-area
    // heuristic over the three axes, restricted to cuts that
    //   - leave a multiple of the leaf size on the left, so that leaves come out full (a plain
    //     median split of 10 000 spheres fills its 4-slot leaves to 65 %: 3856 leaves instead of
    //     2500, and every leaf visit tests 4 slots whether they are used or not);
    //   - keep both sides between 3/8 and 5/8 of the range, which bounds the depth (the traversal
    //     stack lives in LDS, one entry per level) at little cost in tree quality.
    // Cost of a cut = area(left) * leaves(left) + area(right) * leaves(right).  Ranges above
    // kSweepMax items are cut at the (rounded) median of the widest axis instead: the sweep sorts
    // the range three times.
    size_t choose_split(size_t lo, size_t hi, const float cmin[3], const float cmax[3])
    {
        const size_t n = hi - lo, L = kBvhLeafSize;
        size_t k_mid = ((n / 2 + L / 2) / L) * L;
        if (k_mid < L) k_mid = L;
        if (k_mid >= n) k_mid = n - 1;
        int widest = 0;
        for (int a = 1; a < 3; ++a)
            if (cmax[a] - cmin[a] > cmax[widest] - cmin[widest]) widest = a;
        constexpr size_t kSweepMax = 1u << 15;
        if (n > kSweepMax) {
            std::nth_element(items.begin() + lo, items.begin() + lo + k_mid, items.begin() + hi,
                             [widest](const Item &x, const Item &y) {
                                 return x.c[widest] < y.c[widest] || (x.c[widest] == y.c[widest] && x.index < y.index);
                             });
            return lo + k_mid;
        }
        size_t k_lo = ((n * 3 / 8 + L - 1) / L) * L, k_hi = (n - n * 3 / 8) / L * L;
        if (k_lo < L) k_lo = L;
        if (k_hi >= n) k_hi = (n - 1) / L * L;
        if (k_lo > k_hi) { k_lo = k_hi = k_mid; }
        float best_cost = INFINITY;
        int best_axis = widest;
        size_t best_k = k_mid;
        std::vector<float> right_area(n + 1);
        for (int axis = 0; axis < 3; ++axis) {
            sort_axis(lo, hi, axis);
            float blo[3] = {INFINITY, INFINITY, INFINITY}, bhi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (size_t i = n; i-- > 0;) {
                const Item &it = items[lo + i];
                for (int a = 0; a < 3; ++a) { blo[a] = std::min(blo[a], it.c[a] - it.R); bhi[a] = std::max(bhi[a], it.c[a] + it.R); }
                right_area[i] = half_area(blo, bhi);
            }
            for (int a = 0; a < 3; ++a) { blo[a] = INFINITY; bhi[a] = -INFINITY; }
            for (size_t i = 0; i < n; ++i) {
                const Item &it = items[lo + i];
                for (int a = 0; a < 3; ++a) { blo[a] = std::min(blo[a], it.c[a] - it.R); bhi[a] = std::max(bhi[a], it.c[a] + it.R); }
                const size_t k = i + 1;   // items on the left
                if (k < k_lo || k > k_hi || (k % L != 0 && k != k_mid)) continue;
                const float cost = half_area(blo, bhi) * (float)((k + L - 1) / L) + right_area[k] * (float)((n - k + L - 1) / L);
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_k = k; }
            }
        }
        if (best_axis != 2) sort_axis(lo, hi, best_axis);
        return lo + best_k;
    }

    Subtree build(size_t lo, size_t hi)
    {
        Subtree me;
        float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int a = 0; a < 3; ++a) { me.lo[a] = INFINITY; me.hi[a] = -INFINITY; }
        for (size_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                me.lo[a] = std::min(me.lo[a], items[i].c[a] - items[i].R);
                me.hi[a] = std::max(me.hi[a], items[i].c[a] + items[i].R);
                cmin[a] = std::min(cmin[a], items[i].c[a]);
                cmax[a] = std::max(cmax[a], items[i].c[a]);
            }
        if (hi - lo <= kBvhLeafSize) {
            const int leaf = (int)(out->leaf_index.size() / kBvhLeafSize);
            for (size_t k = 0; k < kBvhLeafSize; ++k) {
                if (lo + k < hi) {
                    out->leaf_sph.push_back(sph[items[lo + k].index]);
                    out->leaf_index.push_back(items[lo + k].index);
                } else {
                    // r*r = -inf: det is -inf or NaN for every ray, the test can never succeed
                    out->leaf_sph.push_back(make_float4(0.f, 0.f, 0.f, -INFINITY));
                    out->leaf_index.push_back(0x7fffffff);
                }
            }
            me.ref = ~leaf;
            return me;
        }
        const int node = (int)binary.size();
        binary.emplace_back();
        const size_t mid = choose_split(lo, hi, cmin, cmax);
        const Subtree c0 = build(lo, mid);
        const Subtree c1 = build(mid, hi);
        binary[(size_t)node].child[0] = c0;
        binary[(size_t)node].child[1] = c1;
        me.ref = node;
        return me;
    }

    // Folds the binary tree into the four-wide one: the children of a wide node are the binary
    // node's two children, with the inner child of the largest surface area replaced by its own
    // two children until four are reached (or only leaves are left).  Returns the number of wide
    // levels below and including this node.
    uint32_t emit_wide(int binary_node, int wide_node)
    {
        std::vector<Subtree> kids = {binary[(size_t)binary_node].child[0], binary[(size_t)binary_node].child[1]};
        while (kids.size() < kBvhWidth) {
            int pick = -1;
            float pick_area = -1.f;
            for (size_t k = 0; k < kids.size(); ++k)
                if (kids[k].ref >= 0 && half_area(kids[k].lo, kids[k].hi) > pick_area) {
                    pick = (int)k;
                    pick_area = half_area(kids[k].lo, kids[k].hi);
                }
            if (pick < 0) break;
            const BinaryNode opened = binary[(size_t)kids[(size_t)pick].ref];
            kids[(size_t)pick] = opened.child[0];
            kids.insert(kids.begin() + pick + 1, opened.child[1]);
        }
        // reserve the wide nodes of the inner children first: a node's index must be known before its subtree is emitted
        int refs[kBvhWidth];
        for (size_t k = 0; k < kBvhWidth; ++k) {
            refs[k] = kBvhNoChild;
            if (k < kids.size()) {
                refs[k] = kids[k].ref;
                if (kids[k].ref >= 0) {
                    refs[k] = (int)(out->nodes.size() / kBvhNodeVec4);
                    out->nodes.resize(out->nodes.size() + kBvhNodeVec4);
                }
            }
        }
        float v[kBvhNodeVec4][4];
        for (size_t k = 0; k < kBvhWidth; ++k) {
            const bool used = k < kids.size();
            for (int a = 0; a < 3; ++a) {
                v[a][k] = used ? kids[k].lo[a] : NAN;
                v[3 + a][k] = used ? kids[k].hi[a] : NAN;
            }
            memcpy(&v[6][k], &refs[k], 4);
            v[7][k] = 0.f;
        }
        for (size_t q = 0; q < kBvhNodeVec4; ++q)
            out->nodes[(size_t)wide_node * kBvhNodeVec4 + q] = make_float4(v[q][0], v[q][1], v[q][2], v[q][3]);
        uint32_t below = 0;
        for (size_t k = 0; k < kids.size(); ++k)
            if (kids[k].ref >= 0) below = std::max(below, emit_wide(kids[k].ref, refs[k]));
        return 1 + below;
    }
};

} // namespace

// The 16-bit grid form of the nodes (layout: pt_bvh.hpp).  lo / hi: extent of the bounded spheres;
// box_slack: by how much a node box may exceed that extent on either side (the builder's slack on
// every radius).  Returns false when a plane of a used box falls off the grid, whose clamped
// coordinate would make the box tighter than the traversal's error bound assumes: the scene is
// then rendered by the linear scan.
static bool quantize_nodes(BvhData *out, const float lo[3], const float hi[3], double box_slack)
{
    const size_t n_nodes = out->nodes.size() / kBvhNodeVec4;
    double origin[3], step[3];
    for (int a = 0; a < 3; ++a) {
        // the node boxes exceed the spheres' extent by their slack: leave a margin of 1/64 of the
        // extent on both sides -- or the slack itself where that is more, as on the short axis of a
        // flat or elongated cloud, whose slack follows the cloud's diagonal -- and a little, and
        // four grid steps for the outward rounding
        const double extent = std::max((double)hi[a] - (double)lo[a], 1e-3);
        const double margin = std::max(extent / 64, box_slack) + 1e-3;
        step[a] = (extent + 2 * margin) / 65520.0;
        out->grid_step[a] = (float)step[a];
        step[a] = (double)out->grid_step[a];                      // the step the device uses
        out->grid_origin[a] = (float)((double)lo[a] - margin - 4 * step[a]);
        origin[a] = (double)out->grid_origin[a];
    }
    out->qnodes.assign(n_nodes * 4, make_uint4(0u, 0u, 0u, 0u));
    bool on_grid = true;
    for (size_t k = 0; k < n_nodes; ++k) {
        const float4 *N = &out->nodes[k * kBvhNodeVec4];
        uint32_t q[6][4];
        for (int plane = 0; plane < 6; ++plane) {
            const float v[4] = {N[plane].x, N[plane].y, N[plane].z, N[plane].w};
            const int a = plane % 3;
            for (int c = 0; c < 4; ++c) {
                double g = ((double)v[c] - origin[a]) / step[a];
                g = plane < 3 ? std::floor(g) - 2 : std::ceil(g) + 2;
                if (g < 0 || g > 65535) on_grid = false;           // a used slot's plane (NaN compares false)
                if (!(g >= 0)) g = 0;                              // NaN (unused slot) as well
                if (g > 65535) g = 65535;
                q[plane][c] = (uint32_t)g;
            }
        }
        auto pack = [&q](int plane, int pair) { return q[plane][2 * pair] | (q[plane][2 * pair + 1] << 16); };
        uint4 *Q = &out->qnodes[k * 4];
        Q[0] = make_uint4(pack(0, 0), pack(0, 1), pack(1, 0), pack(1, 1));
        Q[1] = make_uint4(pack(2, 0), pack(2, 1), pack(3, 0), pack(3, 1));
        Q[2] = make_uint4(pack(4, 0), pack(4, 1), pack(5, 0), pack(5, 1));
        uint32_t r[4];
        memcpy(r, &N[6], 16);
        Q[3] = make_uint4(r[0], r[1], r[2], r[3]);
    }
    return on_grid;
}

bool build_bvh(const float4 *sph, const float *radius, uint32_t n, BvhData *out)
{
    out->nodes.clear(); out->qnodes.clear(); out->leaf_sph.clear(); out->leaf_index.clear();
    out->big_sph.clear(); out->big_index.clear();
    out->depth = 0;
    if (n == 0) return false;

    // spheres much larger than the typical one cannot be bounded usefully: always-tested list
    // (the sphere test only sees r*r, so a negative radius acts like its magnitude)
    std::vector<float> rad(n);
    for (uint32_t i = 0; i < n; ++i) rad[i] = std::fabs(radius[i]);
    radius = rad.data();
    std::vector<float> radii(rad);
    std::nth_element(radii.begin(), radii.begin() + (n * 3) / 4, radii.end());
    const float big_threshold = 8.f * radii[(n * 3) / 4];
    std::vector<uint32_t> small;
    for (uint32_t i = 0; i < n; ++i) {
        const bool finite = std::isfinite(radius[i]) && std::isfinite(sph[i].x) && std::isfinite(sph[i].y) &&
                            std::isfinite(sph[i].z);
        if (!finite || !(radius[i] <= big_threshold)) {
            out->big_sph.push_back(sph[i]);
            out->big_index.push_back((int)i);
        } else {
            small.push_back(i);
        }
    }
    if (small.size() <= kBvhLeafSize || out->big_sph.size() > kBvhMaxBig) return false;

    // cloud of the bounded spheres: centre and half-diagonal
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i : small) {
        const float c[3] = {sph[i].x, sph[i].y, sph[i].z};
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], c[a] - radius[i]);
            hi[a] = std::max(hi[a], c[a] + radius[i]);
        }
    }
    double hd2 = 0;
    for (int a = 0; a < 3; ++a) {
        out->centre[a] = 0.5f * (lo[a] + hi[a]);
        hd2 += 0.25 * (double)(hi[a] - lo[a]) * (double)(hi[a] - lo[a]);
    }
    out->half_diag = (float)(std::sqrt(hd2) * (1.0 + 1e-6));
    out->r_min = INFINITY;
    const double slack = 1e-6 * std::sqrt(hd2) + 1e-4;

    Builder b;
    b.sph = sph;
    b.out = out;
    b.items.reserve(small.size());
    for (uint32_t i : small) {
        Item it;
        it.c[0] = sph[i].x; it.c[1] = sph[i].y; it.c[2] = sph[i].z;
        it.R = (float)(((double)radius[i] + slack) * (1.0 + 1e-6));
        it.index = (int)i;
        b.items.push_back(it);
        out->r_min = std::min(out->r_min, radius[i]);
    }
    const Subtree root = b.build(0, b.items.size());   // more than one leaf: the root is an inner node
    out->nodes.resize(kBvhNodeVec4);                   // wide node 0
    out->depth = b.emit_wide(root.ref, 0);
    // by how much an item's box radius exceeds the sphere's: (r + slack) * (1 + 1e-6) - r
    double box_slack = 0;
    for (const Item &it : b.items) box_slack = std::max(box_slack, (double)it.R - (double)radius[it.index]);
    if (!quantize_nodes(out, lo, hi, box_slack)) return false;
    return out->depth <= kBvhMaxDepth;
}

} // namespace ptmi
