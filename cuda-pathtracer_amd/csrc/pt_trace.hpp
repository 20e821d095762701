// Pieces of the trace loop shared by the megakernel (pt_kernels.hip) and the wavefront kernels
// (pt_wavefront.hip): scene staging in LDS, the nearest-hit scan, camera-ray generation and the
// shading of one bounce.  Both kernel families call the same functions, so their results are
// bit-identical by construction.
#pragma once

#include "pt_device.hpp"
#include "pt_kernels.hpp"

namespace ptmi {

constexpr int kBlockThreads = 256;

// Wave-wide sum of a 32-bit counter (wave64), result valid in every lane.
PT_DEV uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Value of `x` in lane `src` (wave-uniform) for every lane.
PT_DEV float wave_read(float x, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), src));
}

// Bytes of LDS stage_scene<false>() fills, rounded up to 16 (device-side twin of scene_lds_bytes).
PT_DEV uint32_t scene_lds_bytes_dev(const SceneParams &s)
{
    const uint32_t n = s.n_prims * 32u + s.n_mats * 24u + s.n_lights * 16u + s.n_tris * 64u;
    return (n + 15u) & ~15u;
}

struct SceneView {
    const float4 *sph;   // LDS: all spheres (resident) or the current tile (streaming)
    const float4 *aux;   // (1/radius, materialId bits, lightId bits, radius)
    const float4 *mat;   // (r, g, b, coef)
    const int2 *mat_type;   // (MaterialType, bits of 1.f / coef)
    const float4 *light; // (r, g, b, -)
    const float4 *tri;   // resident scenes with triangles: four float4 per triangle in LDS (else device memory)
};

// Resident scenes: copy spheres, aux, materials and lights into LDS (one barrier inside).
// Streaming scenes: only the sphere tile lives in LDS, the rest is read from global memory.
template <bool TILED>
PT_DEV SceneView stage_scene(const SceneParams &p, unsigned char *lds_raw, int tid)
{
    SceneView sc;
    float4 *lds_sph = reinterpret_cast<float4 *>(lds_raw);
    if (!TILED) {
        float4 *lds_aux = lds_sph + p.n_prims;
        float4 *lds_mat = lds_aux + p.n_prims;
        float4 *lds_light = lds_mat + p.n_mats;
        int2 *lds_type = reinterpret_cast<int2 *>(lds_light + p.n_lights);
        for (uint32_t i = tid; i < p.n_prims; i += kBlockThreads) {
            lds_sph[i] = p.sph[i];
            lds_aux[i] = p.aux[i];
        }
        for (uint32_t i = tid; i < p.n_mats; i += kBlockThreads) {
            lds_mat[i] = p.mat[i];
            lds_type[i] = p.mat_type[i];
        }
        for (uint32_t i = tid; i < p.n_lights; i += kBlockThreads) lds_light[i] = p.light[i];
        // triangles behind the 8-byte type entries, on the next 16-byte boundary
        float4 *lds_tri = reinterpret_cast<float4 *>(lds_raw + ((p.n_prims * 32u + p.n_mats * 24u + p.n_lights * 16u + 15u) & ~15u));
        for (uint32_t i = tid; i < p.n_tris * 4u; i += kBlockThreads) lds_tri[i] = p.tri[i];
        __syncthreads();
        sc.sph = lds_sph; sc.aux = lds_aux; sc.mat = lds_mat; sc.mat_type = lds_type; sc.light = lds_light;
        sc.tri = lds_tri;
    } else {
        sc.sph = lds_sph; sc.aux = p.aux; sc.mat = p.mat; sc.mat_type = p.mat_type; sc.light = p.light;
        sc.tri = p.tri;
    }
    return sc;
}

// Scan `count` spheres that sit in LDS starting at global index `base` (index order).
PT_DEV void scan_lds(const float4 *sph, int base, int count, V3 o, V3 d, float &tmax, int &best)
{
    // Scenes of up to 16 spheres (the demo scene has 9): the same groups of four, unrolled, so that a
    // sphere's index is an immediate of the select that records the hit instead of a register that has
    // to be loaded from a loop counter first, and the loop's own counting goes away.
    if (base == 0 && count <= 16) {
#pragma unroll
        for (int g = 0; g < 16; g += 4) {
            if (g + 4 <= count) {
                const float4 s0 = sph[g], s1 = sph[g + 1], s2 = sph[g + 2], s3 = sph[g + 3];
                sphere_test(s0, g, o, d, tmax, best);
                sphere_test(s1, g + 1, o, d, tmax, best);
                sphere_test(s2, g + 2, o, d, tmax, best);
                sphere_test(s3, g + 3, o, d, tmax, best);
            } else {
                if (g < count) sphere_test(sph[g], g, o, d, tmax, best);
                if (g + 1 < count) sphere_test(sph[g + 1], g + 1, o, d, tmax, best);
                if (g + 2 < count) sphere_test(sph[g + 2], g + 2, o, d, tmax, best);
                break;
            }
        }
        return;
    }
    int i = 0;
    for (; i + 4 <= count; i += 4) {
        const float4 s0 = sph[i], s1 = sph[i + 1], s2 = sph[i + 2], s3 = sph[i + 3];
        sphere_test(s0, base + i, o, d, tmax, best);
        sphere_test(s1, base + i + 1, o, d, tmax, best);
        sphere_test(s2, base + i + 2, o, d, tmax, best);
        sphere_test(s3, base + i + 3, o, d, tmax, best);
    }
    for (; i < count; ++i) sphere_test(sph[i], base + i, o, d, tmax, best);
}

// Second half of sphere_test (primitive.h:43-44, scene.h:78-80) for a discriminant already computed.
PT_DEV void sphere_accept(float b, float det, int index, float &tmax, int &best)
{
    if (!(det < 0.f)) {
        const float sq = pt_sqrt_discriminant(det);
        float t = b - sq;
        if (!(t > PT_EPSILON)) t = b + sq;
        if (t > PT_EPSILON && t < tmax) { tmax = t; best = index; }
    }
}

// scan_lds for scenes of thousands of spheres, where a wave's rays meet few of them (on the
// 10 007-sphere scene 2 % of the spheres have a non-negative discriminant for any lane): the first
// halves of eight sphere tests run back to back, and ONE test of the eight sign bits decides whether
// any second half is needed -- an eighth of the compares and branches of the plain loop (groups of
// 4: +2.5 %, of 8: +6 %, of 16: no more).  The
// second halves, when needed, run in index order.  (A NaN discriminant with its sign bit set is
// skipped here and rejected there: no difference in the result.)
PT_DEV void scan_lds_sparse(const float4 *sph, int base, int count, V3 o, V3 d, float &tmax, int &best)
{
    int i = 0;
    for (; i + 8 <= count; i += 8) {
        float b[8], det[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 s = sph[i + k];
            const float opx = s.x - o.x, opy = s.y - o.y, opz = s.z - o.z;
            b[k] = opx * d.x + opy * d.y + opz * d.z;
            det[k] = b[k] * b[k] - (opx * opx + opy * opy + opz * opz) + s.w;
        }
        uint32_t all_negative = __float_as_uint(det[0]);
#pragma unroll
        for (int k = 1; k < 8; ++k) all_negative &= __float_as_uint(det[k]);
        if (!(all_negative & 0x80000000u)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) sphere_accept(b[k], det[k], base + i + k, tmax, best);
        }
    }
    for (; i < count; ++i) sphere_test(sph[i], base + i, o, d, tmax, best);
}

// The sphere test with the order-independent acceptance rule of the BVH search: the brute-force
// scan keeps the lowest index among equal distances because it visits spheres in index order;
// visiting them in any order, (t, index) must be compared lexicographically to get the same.
PT_DEV void sphere_test_lex(float4 s, int index, V3 o, V3 d, float &tmax, int &best)
{
    const float opx = s.x - o.x, opy = s.y - o.y, opz = s.z - o.z;
    const float b = opx * d.x + opy * d.y + opz * d.z;
    const float det = b * b - (opx * opx + opy * opy + opz * opz) + s.w;
    if (!(det < 0.f)) {
        const float sq = pt_sqrt_discriminant(det);
        float t = b - sq;
        if (!(t > PT_EPSILON)) t = b + sq;
        if (t > PT_EPSILON && (t < tmax || (t == tmax && best >= 0 && index < best))) { tmax = t; best = index; }
    }
}

// The ray as the slab tests see it.  A box plane is stored as a 16-bit grid coordinate q (plane =
// origin + q * step) and reaches the test as the float 2^23 + q (its bits are 0x4b000000 | q: no
// conversion instruction).  The boxes are widened per ray by w(t) = a + s * t, growing along the ray
// (see bvh_begin).  On one axis the point o + t * d is inside the widened slab when
//     t * (d + s) >= lo - a - o   and   t * (d - s) <= hi + a - o,
// so the two planes get different slopes: t(lower plane) = (2^23 + q) * inv_lo + klo with inv_lo =
// step / (d + s), klo = (origin - a - o) / (d + s) - 2^23 * inv_lo, and t(upper plane) likewise with
// d - s and + a.  For d > s the lower plane's t bounds t from below (the ray enters there) and the
// upper plane's from above (it leaves there); for d < -s the divisions by negative numbers turn
// both around.  For |d| <= s the slab grows at least as fast as the ray moves across it: both
// conditions are lower bounds on t, and the one of the plane the ray moves away from is dropped
// (its t is replaced by 3e38: "never leaves").  In every case the plane of entry is the lower one
// where the direction component is >= 0 and the upper one where it is negative, so the two are told
// apart once per ray (`neg`, bit k for axis k) and a box costs one max3 and one min3 instead of a
// min and a max per axis: (inv_n, kn) belong to the plane of entry, (inv_f, kf) to the plane of
// exit.  The fused multiply-add rounds once; what k loses to the large constant is below half a
// grid step, which the builder's outward rounding covers.
struct BvhRay {
    V3 inv_n, inv_f, kn, kf;
    uint32_t neg;
};

// The two 16-bit grid coordinates of a node word as the floats 2^23 + q.
PT_DEV float qplane_lo(uint32_t w) { return __uint_as_float((w & 0xffffu) | 0x4b000000u); }
PT_DEV float qplane_hi(uint32_t w) { return __uint_as_float(__builtin_amdgcn_alignbit(0x4b00u, w, 16u)); }

// Slab test of one child box against the ray.  The fused multiply-adds are deliberate: this
// arithmetic only decides which boxes are visited (its rounding is covered by the widening),
// it never reaches a result.  NaN-conservative: the negated comparisons enter on NaN.
// (nx, ny, nz): the planes of entry, (fx, fy, fz): the planes of exit.
PT_DEV bool slab_test(float nx, float ny, float nz, float fx, float fy, float fz, const BvhRay &r, float tmax,
                      float &tnear)
{
    const float t1x = __builtin_fmaf(nx, r.inv_n.x, r.kn.x), t2x = __builtin_fmaf(fx, r.inv_f.x, r.kf.x);
    const float t1y = __builtin_fmaf(ny, r.inv_n.y, r.kn.y), t2y = __builtin_fmaf(fy, r.inv_f.y, r.kf.y);
    const float t1z = __builtin_fmaf(nz, r.inv_n.z, r.kn.z), t2z = __builtin_fmaf(fz, r.inv_f.z, r.kf.z);
    tnear = __builtin_fmaxf(__builtin_fmaxf(t1x, t1y), t1z);
    const float tfar = __builtin_fminf(__builtin_fminf(t2x, t2y), t2z);
    return !(tnear > tfar) && !(tfar < 0.f) && !(tnear > tmax);
}

// Per-ray constants of the box widening, see bvh_begin.
struct BvhMargin {
    float eps;     // max(|dir|^2 - 1, 0) + rounding allowance
    float err;     // 1e-6 * L^2 * |dir|^2: rounding error of the sphere test's discriminant
    float slack;   // rounding of the slab arithmetic
    float r_min;
    // sqrt(r_min^2 + E2) - r_min, rounded up: hardware square root and reciprocal (1 ulp each) with
    // a factor that covers their errors -- this arithmetic only decides which boxes are visited
    PT_DEV float at(float T) const
    {
        const float E2 = eps * T * T + err;
        return E2 * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(r_min * r_min + E2) * 0.999999f + r_min) * 1.001f + slack;
    }
};

// 1 / x for the slab tests, finite for every finite or zero x (a zero component becomes a tiny
// one: the ray then leaves the scene's extent along that axis only after ~1e18 units).
PT_DEV float bvh_inverse(float x)
{
    const float a = __builtin_fabsf(x) > 1e-18f ? x : __builtin_copysignf(1e-18f, x);
    return __builtin_amdgcn_rcpf(a);
}

// Per-lane traversal stacks in LDS, entry k of lane l at [k][l] (conflict-free).  An entry is a
// deferred child reference and the distance at which the ray enters its box.
template <int STRIDE>
struct BvhStack8 {
    uint32_t *base;   // this lane's column: entry k is the words [2k][lane] (child) and [2k + 1][lane] (distance),
                      // two conflict-free dword accesses a constant 4 * STRIDE bytes apart (no 64-bit register pair to build)
    PT_DEV void push(int sp, int ref, float t) const { base[2 * sp * STRIDE] = (uint32_t)ref; base[(2 * sp + 1) * STRIDE] = __float_as_uint(t); }
    PT_DEV void read(int sp, int &ref, float &t) const { ref = (int)base[2 * sp * STRIDE]; t = __uint_as_float(base[(2 * sp + 1) * STRIDE]); }
};
// Pops deferred boxes until one can still hold a hit at or below tmax; false when none is left.
template <class Stack>
PT_DEV bool bvh_pop(const Stack &stack, int &sp, float tmax, int &cur)
{
    while (sp > 0) {
        --sp;
        int ref;
        float t;
        stack.read(sp, ref, t);
        if (!(t > tmax)) { cur = ref; return true; }
    }
    return false;
}

// Nearest hit through the sphere BVH (host side: pt_bvh.cpp).  Depth-first, nearer child first,
// the farther one deferred on a per-lane stack in LDS (`stack`, [B.depth][kBlockThreads]) together
// with its entry distance, so that it is dropped when a closer hit has been found meanwhile.
// The loop alternates between "descend until this lane holds a leaf" and "test the leaf", so
// that the lanes of a wave run the box code together and the sphere code together.
// A box is entered when the ray's slab interval through the widened box overlaps [0, tmax].
// The widening is per ray: a valid t of the fp32 sphere test puts origin + t * dir at squared
// distance r^2 + (|dir|^2 - 1) * t^2 + err from the centre, err < 1e-6 * |op|^2 * |dir|^2, op =
// centre - origin (derivation in pt_bvh.cpp).  With eps >= |dir|^2 - 1 and L >= |op| for every
// bounded sphere, the hit point lies within e(t) = sqrt(r_min^2 + eps * t^2 + err(L)) - r_min of
// the sphere's box (the excess shrinks as r grows); the slab arithmetic's own rounding is added on
// top.  Mirror directions are not re-normalised (globals.h:107-109) and can be several per cent too
// long after a bounce off a small sphere, so eps is not always tiny and e grows to tens of units
// at the far end of such a ray.  t itself is below tmax and below T0 = (2 |dir| + 0.1) L: the hit
// lies about r from a centre, and |op| + r <= L.  e(t) is convex (the norm of (sqrt(r_min^2 + err),
// sqrt(eps) t) minus a constant), so on [0, T0] it stays below the chord through its end points:
// the boxes are widened by w(t) = a + s * t with a >= e(0) and a + s * T0 >= e(T0), narrow near the
// origin and wide far away, and BvhRay turns that into two slopes per axis at no cost per box.
// A skipped box therefore cannot hold a sphere whose test would succeed with t <= tmax, and
// sphere_test_lex makes the visiting order irrelevant: the result equals the brute-force scan's.
//
// Visit budget: a few rays (long mirror directions widen their boxes enormously) would visit
// thousands of boxes and hold their whole wave up.  A lane that exceeds B.budget visits stops
// with its partial result -- a valid candidate, the lexicographic minimum does not care about
// order -- and the wave then finishes such rays together, one at a time: 64 lanes scan the sphere
// array in stride and reduce (t, index) across the wave.  Every lane of the wave must call this
// function (`active` masks the lanes that carry a ray).


struct BvhTraversal {
    BvhRay ray;
    int cur, sp;       // current node reference and stack height
    uint32_t visits;
#ifdef PT_BVH_STATS
    bool long_direction;
#endif
};

// Starts a traversal: tests the always-tested spheres, derives the ray's widening and slab
// constants.  Returns false when there is nothing to traverse (NaN in the ray).
PT_DEV bool bvh_begin(const BvhParams &B, V3 o, V3 d, float &tmax, int &best, BvhTraversal &tr)
{
    // best is -1 on entry and big_index ascends: the strict rule of the scan gives the lexicographic minimum
    for (uint32_t i = 0; i < B.n_big; ++i) sphere_test(B.big_sph[i], B.big_index[i], o, d, tmax, best);
    const float cx = o.x - B.centre[0], cy = o.y - B.centre[1], cz = o.z - B.centre[2];
    const float L = __builtin_amdgcn_sqrtf(cx * cx + cy * cy + cz * cz) * 1.000002f + B.half_diag;
    const float excess = (d.x * d.x + d.y * d.y + d.z * d.z) - 1.f;
    BvhMargin m;
    m.eps = __builtin_fmaxf(excess, 0.f) * 1.000001f + 4e-7f;
    // b = op . dir grows with |dir|, and with it the rounding errors of b * b and of t = b -+ sqrt(det)
    m.err = 1e-6f * L * L * (1.f + m.eps);
    // slab arithmetic: plane * inv + k cancels two terms of the size of the coordinates
    m.slack = (2e-6f * L + 4e-7f * (__builtin_fabsf(o.x) + __builtin_fabsf(o.y) + __builtin_fabsf(o.z))) * (1.f + m.eps);
    m.r_min = B.r_min;
    // the chord of e(t) over [0, T0]; e(0) - slack >= 0 makes (e(T0) - slack) / T0 a large enough slope
    const float T0 = __builtin_fminf(tmax, (2.2f * (1.f + m.eps) + 0.1f) * L);
    float a = m.at(0.f);
    float s = (m.at(T0) - m.slack) * __builtin_amdgcn_rcpf(T0) * 1.00001f;
    // overflow (absurdly long directions, far origins, an empty search interval): visit everything, like the scan
    if (!(s < 3.0e38f)) { a = PT_INF; s = 0.f; }
#ifdef PT_BVH_STATS
    tr.long_direction = m.eps > 2e-6f;
#endif
    const float g[3] = {B.grid_origin[0] - o.x, B.grid_origin[1] - o.y, B.grid_origin[2] - o.z};
    const float dir[3] = {d.x, d.y, d.z};
    float il[3], ih[3], kl[3], kh[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const float p = bvh_inverse(dir[ax] + s), q = bvh_inverse(dir[ax] - s);
        il[ax] = p * B.grid_step[ax];
        ih[ax] = q * B.grid_step[ax];
        kl[ax] = (g[ax] - a) * p - 8388608.f * il[ax];
        kh[ax] = (g[ax] + a) * q - 8388608.f * ih[ax];
        if (!(__builtin_fabsf(dir[ax]) > s)) {   // the slab outgrows the ray's motion across it
            if (dir[ax] < 0.f) { il[ax] = 0.f; kl[ax] = 3.0e38f; }
            else               { ih[ax] = 0.f; kh[ax] = 3.0e38f; }
        }
    }
    // plane of entry: the lower one for a direction component >= 0 (its bound is the lower bound on t
    // in the narrow case |d| <= s as well, where the other plane's has been dropped), else the upper one
    float in_[3], if_[3], kn_[3], kf_[3];
    uint32_t neg = 0;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        const bool ng = dir[ax] < 0.f;
        in_[ax] = ng ? ih[ax] : il[ax]; kn_[ax] = ng ? kh[ax] : kl[ax];
        if_[ax] = ng ? il[ax] : ih[ax]; kf_[ax] = ng ? kl[ax] : kh[ax];
        neg |= ng ? (1u << ax) : 0u;
    }
    tr.ray.inv_n = mk(in_[0], in_[1], in_[2]); tr.ray.inv_f = mk(if_[0], if_[1], if_[2]);
    tr.ray.kn = mk(kn_[0], kn_[1], kn_[2]);    tr.ray.kf = mk(kf_[0], kf_[1], kf_[2]);
    tr.ray.neg = neg;
    tr.cur = 0; tr.sp = 0; tr.visits = 0;
    // a NaN in the origin or the direction makes every sphere test fail (all comparisons false)
    return !(excess != excess || L != L);
}

// One inner node (tr.cur >= 0) of the four-wide tree: tests its up to four child boxes, descends
// into the nearest one that is hit and defers the others on the stack (in slot order: the order of
// visits affects only how early the bound tightens, never the result).  Returns false when the
// traversal is finished (stack empty).  A full stack makes the ray give up the tree: its visit
// count is set beyond any budget, so that the caller finishes it by the wave-wide scan.
template <class Stack>
PT_DEV bool bvh_node_step(const BvhParams &B, const uint4 *nodes, const Stack &stack, float tmax, BvhTraversal &tr)
{
    PT_LS(23);
    const uint4 *N = nodes + 4 * (size_t)tr.cur;   // device memory, or the workgroup's copy in LDS
    const uint4 q0 = N[0], q1 = N[1], q2 = N[2], rf = N[3];
    const int c0 = (int)rf.x, c1 = (int)rf.y, c2 = (int)rf.z, c3 = (int)rf.w;
    float t0, t1, t2, t3;
    // words: q0 = (lo.x c0|c1, lo.x c2|c3, lo.y c0|c1, lo.y c2|c3), q1 = (lo.z, lo.z, hi.x, hi.x), q2 = (hi.y, hi.y, hi.z, hi.z)
    // planes of entry and exit per axis: the lower and the upper plane's words, swapped where the ray runs backwards
    const bool gx = (tr.ray.neg & 1u) != 0u, gy = (tr.ray.neg & 2u) != 0u, gz = (tr.ray.neg & 4u) != 0u;
    const uint32_t nx01 = gx ? q1.z : q0.x, nx23 = gx ? q1.w : q0.y, fx01 = gx ? q0.x : q1.z, fx23 = gx ? q0.y : q1.w;
    const uint32_t ny01 = gy ? q2.x : q0.z, ny23 = gy ? q2.y : q0.w, fy01 = gy ? q0.z : q2.x, fy23 = gy ? q0.w : q2.y;
    const uint32_t nz01 = gz ? q2.z : q1.x, nz23 = gz ? q2.w : q1.y, fz01 = gz ? q1.x : q2.z, fz23 = gz ? q1.y : q2.w;
    const bool h0 = slab_test(qplane_lo(nx01), qplane_lo(ny01), qplane_lo(nz01), qplane_lo(fx01), qplane_lo(fy01), qplane_lo(fz01),
                                 tr.ray, tmax, t0);   // slots 0 and 1 are always used
    const bool h1 = slab_test(qplane_hi(nx01), qplane_hi(ny01), qplane_hi(nz01), qplane_hi(fx01), qplane_hi(fy01), qplane_hi(fz01),
                                 tr.ray, tmax, t1);
    const bool h2 = c2 != 0x7fffffff &&
                    slab_test(qplane_lo(nx23), qplane_lo(ny23), qplane_lo(nz23), qplane_lo(fx23), qplane_lo(fy23), qplane_lo(fz23),
                                 tr.ray, tmax, t2);
    const bool h3 = c3 != 0x7fffffff &&
                    slab_test(qplane_hi(nx23), qplane_hi(ny23), qplane_hi(nz23), qplane_hi(fx23), qplane_hi(fy23), qplane_hi(fz23),
                                 tr.ray, tmax, t3);
    ++tr.visits;
    if (!(h0 || h1 || h2 || h3)) return bvh_pop(stack, tr.sp, tmax, tr.cur);
    // The nearest hit child, the lowest slot among equal distances (only a visiting order: a NaN
    // distance would count as nearest).  Some hit child always qualifies: the minimum itself.
    const float e0 = h0 ? t0 : PT_INF, e1 = h1 ? t1 : PT_INF, e2 = h2 ? t2 : PT_INF, e3 = h3 ? t3 : PT_INF;
    const float emin = __builtin_fminf(__builtin_fminf(e0, e1), __builtin_fminf(e2, e3));
    const bool n0 = h0 && !(e0 > emin);
    const bool n1 = h1 && !n0 && !(e1 > emin);
    const bool n2 = h2 && !(n0 || n1) && !(e2 > emin);
    const bool n3 = h3 && !(n0 || n1 || n2);
    const bool d0 = h0 && !n0, d1 = h1 && !n1, d2 = h2 && !n2, d3 = h3 && !n3;   // deferred
    if (tr.sp + 3 > (int)B.depth) {   // near the top of the stack: count what this node adds
        const int n_defer = (int)d0 + (int)d1 + (int)d2 + (int)d3;
        if (tr.sp + n_defer > (int)B.depth) { tr.visits = 0xffffffffu; return true; }
    }
    if (d0) { stack.push(tr.sp, c0, t0); ++tr.sp; }
    if (d1) { stack.push(tr.sp, c1, t1); ++tr.sp; }
    if (d2) { stack.push(tr.sp, c2, t2); ++tr.sp; }
    if (d3) { stack.push(tr.sp, c3, t3); ++tr.sp; }
    tr.cur = n0 ? c0 : (n1 ? c1 : (n2 ? c2 : c3));
    return true;
}

// While a traversal runs, a hit inside a leaf is remembered as the leaf slot it sits in (tagged
// with kLeafSlotTag), not as the sphere's index: the slot's index is needed only for the winner
// and for breaking exact ties, so the leaf step does not load the leaf's four indices (64 instead
// of 80 bytes per leaf visit).  bvh_sphere_index turns either form into the sphere's index.
constexpr int kLeafSlotTag = 0x40000000;
PT_DEV int bvh_sphere_index(const BvhParams &B, int x)
{
    return (x >= 0 && (x & kLeafSlotTag)) ? reinterpret_cast<const int *>(B.leaf_index)[x & ~kLeafSlotTag] : x;
}

// sphere_test_lex for the sphere in leaf slot `slot` (tagged).
PT_DEV void sphere_test_slot(const BvhParams &B, float4 s, int slot, V3 o, V3 d, float &tmax, int &best)
{
    const float opx = s.x - o.x, opy = s.y - o.y, opz = s.z - o.z;
    const float b = opx * d.x + opy * d.y + opz * d.z;
    const float det = b * b - (opx * opx + opy * opy + opz * opz) + s.w;
    if (!(det < 0.f)) {
        const float sq = pt_sqrt_discriminant(det);
        float t = b - sq;
        if (!(t > PT_EPSILON)) t = b + sq;
        if (t > PT_EPSILON) {
            bool take = t < tmax;
            if (t == tmax && best >= 0) take = bvh_sphere_index(B, slot) < bvh_sphere_index(B, best);   // rare
            if (take) { tmax = t; best = slot; }
        }
    }
}

// One leaf (tr.cur < 0): its four sphere slots, then the next deferred box.
template <class Stack>
PT_DEV bool bvh_leaf_step(const BvhParams &B, const Stack &stack, V3 o, V3 d, float &tmax, int &best, BvhTraversal &tr)
{
    PT_LS(24);
    const uint32_t leaf = (uint32_t)~tr.cur;
    const float4 *S = B.leaf_sph + 4 * (size_t)leaf;
    const float4 s0 = S[0], s1 = S[1], s2 = S[2], s3 = S[3];
    const int slot = (int)(leaf * 4u) | kLeafSlotTag;
    sphere_test_slot(B, s0, slot, o, d, tmax, best);
    sphere_test_slot(B, s1, slot + 1, o, d, tmax, best);
    sphere_test_slot(B, s2, slot + 2, o, d, tmax, best);
    sphere_test_slot(B, s3, slot + 3, o, d, tmax, best);
    ++tr.visits;
    return bvh_pop(stack, tr.sp, tmax, tr.cur);
}

// Rays over budget: the wave scans all spheres for each of them in turn (every lane of the wave
// must call this; `over_budget` marks the lanes whose (tmax, best) is a partial result).
PT_DEV void bvh_finish_over_budget(const float4 *sph, uint32_t n_prims, bool over_budget, V3 o, V3 d, float &tmax,
                                   int &best)
{
    unsigned long long pending = __ballot(over_budget);
    const uint32_t lane = __lane_id();
    while (pending) {
        const int src = __ffsll(pending) - 1;
        pending &= pending - 1;
        const V3 so = mk(wave_read(o.x, src), wave_read(o.y, src), wave_read(o.z, src));
        const V3 sd = mk(wave_read(d.x, src), wave_read(d.y, src), wave_read(d.z, src));
        float ct = wave_read(tmax, src);
        int cb = __float_as_int(wave_read(__int_as_float(best), src));
        for (uint32_t i = lane; i < n_prims; i += 64) sphere_test_lex(sph[i], (int)i, so, sd, ct, cb);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float t2 = __shfl_xor(ct, off);
            const int b2 = __shfl_xor(cb, off);
            // a miss carries index -1: as unsigned it loses against every hit at the same t
            if (t2 < ct || (t2 == ct && (uint32_t)b2 < (uint32_t)cb)) { ct = t2; cb = b2; }
        }
        if ((int)lane == src) { tmax = ct; best = cb; }
    }
}

PT_DEV void intersect_bvh(const BvhParams &B, const float4 *sph, uint32_t n_prims, uint2 *stack_lds, int tid,
                          bool active, V3 o, V3 d, float &tmax, int &best)
{
    BvhStack8<kBlockThreads> stack;
    stack.base = reinterpret_cast<uint32_t *>(stack_lds) + tid;
    BvhTraversal tr;
    bool more = false;
    if (active) more = bvh_begin(B, o, d, tmax, best, tr);
    bool over_budget = false;
#ifdef PT_BVH_STATS
    uint32_t st_nodes = 0, st_leaves = 0, st_wave_nodes = 0, st_wave_leaves = 0;
    const bool st_leader = (int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1;
#define PT_BVH_COUNT(x) (++(x))
#define PT_BVH_WAVE_COUNT(x) do { if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1) ++(x); } while (0)
#else
#define PT_BVH_COUNT(x) ((void)0)
#define PT_BVH_WAVE_COUNT(x) ((void)0)
#endif
    while (more) {
#pragma unroll 1
        for (uint32_t k = 0; k < B.node_steps; ++k) {
            if (more && tr.cur >= 0) {
                PT_BVH_COUNT(st_nodes);
                PT_BVH_WAVE_COUNT(st_wave_nodes);
                more = bvh_node_step(B, B.nodes, stack, tmax, tr);
                if (tr.visits > B.budget) { over_budget = more; more = false; }
            }
        }
        if (more && tr.cur < 0) {
            PT_BVH_COUNT(st_leaves);
            PT_BVH_WAVE_COUNT(st_wave_leaves);
            more = bvh_leaf_step(B, stack, o, d, tmax, best, tr);
        }
    }
    best = bvh_sphere_index(B, best);
    bvh_finish_over_budget(sph, n_prims, over_budget, o, d, tmax, best);
#ifdef PT_BVH_STATS
    if (active) {
        atomicAdd(B.stats + 0, (unsigned long long)st_nodes);
        atomicAdd(B.stats + 1, (unsigned long long)st_leaves);
        atomicAdd(B.stats + 2, 1ull);
        if (over_budget) atomicAdd(B.stats + 3, 1ull);
        atomicAdd(B.stats + 4, (unsigned long long)st_wave_nodes);
        atomicAdd(B.stats + 5, (unsigned long long)st_wave_leaves);
        if (tr.long_direction) atomicAdd(B.stats + 7, 1ull);
        atomicAdd(B.stats + 8 + (31 - __clz((int)(st_nodes + st_leaves + 1))), 1ull);   // log2 histogram of steps
    }
    if (st_leader) atomicAdd(B.stats + 6, 1ull);
#endif
}

// One triangle of the scan, in the part of Primitive::Intersect (primitive.h:39-45).  No counterpart in
// the reference, whose only primitive is the sphere (primitive.h:25-28).  Moeller-Trumbore in fp32, every
// operation in the order the CPU oracle performs it (no fused multiply-add, correctly rounded 1/det), both
// sides of the triangle.  t0 = (v0, -), t1 = (v1 - v0, -), t2 = (v2 - v0, -).  A degenerate triangle or a ray
// in its plane has det == 0: 1/det is infinite, u infinite or NaN, and the first test fails like every NaN.
PT_DEV void triangle_test(float4 t0, float4 t1, float4 t2, int index, V3 o, V3 d, float &tmax, int &best)
{
    const V3 e1 = mk(t1.x, t1.y, t1.z), e2 = mk(t2.x, t2.y, t2.z);
    const V3 p = cross(d, e2);
    const float det = dot(e1, p);
    const float inv = pt_rcp(det);
    const V3 s = o - mk(t0.x, t0.y, t0.z);
    const float u = dot(s, p) * inv;
    if (u >= 0.f && u <= 1.f) {
        const V3 q = cross(s, e1);
        const float v = dot(d, q) * inv;
        if (v >= 0.f && u + v <= 1.f) {
            const float t = dot(e2, q) * inv;
            if (t > PT_EPSILON && t < tmax) { tmax = t; best = index; }
        }
    }
}

// The triangles of a scene behind its spheres (indices first_index ...), `tri` holding `stride` float4 per
// triangle of which the first three are read.
PT_DEV void scan_triangles(const float4 *tri, int stride, int first_index, int count, V3 o, V3 d, float &tmax, int &best)
{
    for (int k = 0; k < count; ++k) {
        const float4 *t = tri + (size_t)k * stride;
        triangle_test(t[0], t[1], t[2], first_index + k, o, d, tmax, best);
    }
}

// Intersect (scene.h:71-94).  With TILED every thread of the workgroup must call it (barriers in
// the scan, wave-wide cooperation in the BVH search); `active` masks the lanes that carry a ray.
// TRI: the scene's triangles continue the scan behind the spheres (linear scan only).
template <bool TILED, bool BVH = false, bool TRI = false>
PT_DEV void intersect_scene(const SceneParams &p, const SceneView &sc, unsigned char *lds_raw, int tid,
                            bool active, V3 o, V3 d, float &tmax, int &best, float tmax0 = PT_INF)
{
    tmax = tmax0;   // upper bound of the search (tmax of scene.h:72); a hit must be strictly closer
    best = -1;
    if (BVH) {
        intersect_bvh(p.bvh, p.sph, p.n_prims, reinterpret_cast<uint2 *>(lds_raw), tid, active, o, d, tmax, best);
    } else if (TILED) {
        float4 *lds_sph = reinterpret_cast<float4 *>(lds_raw);
        for (uint32_t base = 0; base < p.n_prims; base += kTileSpheres) {
            const uint32_t count = min(kTileSpheres, p.n_prims - base);
            __syncthreads();
            for (uint32_t i = tid; i < count; i += kBlockThreads) lds_sph[i] = p.sph[base + i];
            __syncthreads();
            if (active) scan_lds_sparse(lds_sph, (int)base, (int)count, o, d, tmax, best);
        }
        if (TRI) {   // the triangles pass through the same LDS tile, three float4 each
            for (uint32_t base = 0; base < p.n_tris; base += kTileTriangles) {
                const uint32_t count = min(kTileTriangles, p.n_tris - base);
                __syncthreads();
                for (uint32_t i = tid; i < count * 3u; i += kBlockThreads) {
                    const uint32_t k = i / 3u;
                    lds_sph[i] = p.tri[(size_t)(base + k) * 4u + (i - k * 3u)];
                }
                __syncthreads();
                if (active) scan_triangles(lds_sph, 3, (int)(p.n_prims + base), (int)count, o, d, tmax, best);
            }
        }
    } else {
        if (active) scan_lds(sc.sph, 0, (int)p.n_prims, o, d, tmax, best);
        if (TRI) {
            if (active) scan_triangles(sc.tri, 4, (int)p.n_prims, (int)p.n_tris, o, d, tmax, best);
        }
    }
}

// CameraSamples (pathtracer.cu:34-48) + Camera::GenerateRay (camera.h:66-72) for sub-sample
// `sub` of pixel (x, y), iteration `it`.  Returns the normalised direction.
PT_DEV V3 camera_ray(const CameraParams &cam, uint32_t x, uint32_t y, uint32_t pix, uint32_t it, int sub,
                     uint32_t k0, uint32_t k1)
{
    uint32_t r[4];
    philox4x32_10(pix, it, (uint32_t)(sub >> 1), STREAM_CAMERA, k0, k1, r);
    const float ju = u01((sub & 1) ? r[2] : r[0]) * ((sub & 1) ? 0.5f : -0.5f);
    const float jv = u01((sub & 1) ? r[3] : r[1]) * ((sub & 2) ? 0.5f : -0.5f);
    const float sx = (float)(int)x + ju;
    const float sy = (float)(int)y + jv;
    const V3 first = mk(cam.first[0], cam.first[1], cam.first[2]);
    const V3 pxx = mk(cam.pxx[0], cam.pxx[1], cam.pxx[2]);
    const V3 pxy = mk(cam.pxy[0], cam.pxy[1], cam.pxy[2]);
    return normalize((first - pxy * sy) + pxx * sx);
}

// Two camera rays from one Philox block: sub-samples 2*half and 2*half + 1 of a pixel-iteration,
// the same values camera_ray() returns for them.
PT_DEV void camera_ray_pair(const CameraParams &cam, uint32_t x, uint32_t y, uint32_t pix, uint32_t it,
                            uint32_t half, uint32_t k0, uint32_t k1, V3 dirs[2])
{
    const V3 first = mk(cam.first[0], cam.first[1], cam.first[2]);
    const V3 pxx = mk(cam.pxx[0], cam.pxx[1], cam.pxx[2]);
    const V3 pxy = mk(cam.pxy[0], cam.pxy[1], cam.pxy[2]);
    const float fx = (float)(int)x, fy = (float)(int)y;
    uint32_t r[4];
    philox4x32_10(pix, it, half, STREAM_CAMERA, k0, k1, r);
    const float vs = half ? 0.5f : -0.5f;
    const float sx0 = fx + u01(r[0]) * -0.5f, sy0 = fy + u01(r[1]) * vs;
    const float sx1 = fx + u01(r[2]) * 0.5f, sy1 = fy + u01(r[3]) * vs;
    dirs[0] = normalize((first - pxy * sy0) + pxx * sx0);
    dirs[1] = normalize((first - pxy * sy1) + pxx * sx1);
}

// The four camera rays of one pixel-iteration at once: the same values camera_ray() returns for
// sub = 0..3 (Philox slot 0 feeds sub-samples 0 and 1, slot 1 feeds 2 and 3), with two Philox
// blocks instead of four.
PT_DEV void camera_rays4(const CameraParams &cam, uint32_t x, uint32_t y, uint32_t pix, uint32_t it,
                         uint32_t k0, uint32_t k1, V3 dirs[4])
{
    camera_ray_pair(cam, x, y, pix, it, 0u, k0, k1, dirs);
    camera_ray_pair(cam, x, y, pix, it, 1u, k0, k1, dirs + 2);
}

// One iteration of the loop in Trace (pathtracer.cu:125-168) for a ray that hit sphere `best`
// at distance t: emission, BSDF sample, throughput, Russian roulette, depth limit.  `s` is the
// hit sphere's (cx, cy, cz, r*r).  Updates o, d, T, L and increments b when the path goes on;
// returns true when the path ends -- o, d and T are then left in whatever state the exit found
// them in (a path that has ended has no use for them), which spares the copies that keeping
// their old values across the exits would cost.
//
// shade_surface is the part behind the hit point and the normal: o is the hit point already, n the normal
// of the primitive there, mat_id / light_id its material and light.
PT_DEV bool shade_surface(const SceneView &sc, V3 n, uint32_t mat_id, int light_id, uint32_t pix, uint32_t it,
                          uint32_t k0, uint32_t k1, int mb, V3 &o, V3 &d, V3 &T, V3 &L, int &b)
{
    const V3 wo = d;

    // emission on every light hit: pathtracer.cu:134-137, Light::L light.h:43-45
    PT_LS(5);
    if (light_id > -1) {
        PT_LS(6);
        const float4 li = sc.light[light_id];
        const V3 le = dot(n, -wo) > 0.f ? mk(li.x, li.y, li.z) : mk(0.f, 0.f, 0.f);
        L = L + T * le;
    }

    uint32_t r[4];
    philox4x32_10(pix, it, (uint32_t)b, STREAM_PATH, k0, k1, r);
    const float4 m = sc.mat[mat_id];
    const int2 mt = sc.mat_type[mat_id];
    float pdf;
    const V3 f = sample_material(mt.x, mk(m.x, m.y, m.z), m.w, __int_as_float(mt.y), wo, d, pdf, n, u01(r[0]), u01(r[1]));
    // pathtracer.cu:146-161
    if ((f.x == 0.f && f.y == 0.f && f.z == 0.f) || pdf == 0.f) return true;
    PT_LS(7);
    T = T * vdiv(f * __builtin_fabsf(dot(d, n)), pdf);
    bool end = false;
    if (b > 3) {
        PT_LS(8);
        const float gb = T.y > T.z ? T.y : T.z;   // Color::Max, geometry.h:474
        const float mx = T.x > gb ? T.x : gb;
        const float q = .5f < mx ? .5f : mx;      // min macro, globals.h:66
        if (u01(r[2]) > q) end = true;
        else T = vdiv(T, q);
    }
    if (b == mb) end = true;
    ++b;
    return end;
}

PT_DEV bool shade_bounce(const SceneView &sc, float4 s, int best, float t, uint32_t pix, uint32_t it,
                         uint32_t k0, uint32_t k1, int mb, V3 &o, V3 &d, V3 &T, V3 &L, int &b)
{
    // hit point and normal: scene.h:86-87, primitive.h:74
    const float4 a = sc.aux[best];
    o = o + d * t;
    const V3 n = (o - mk(s.x, s.y, s.z)) * a.x;
    return shade_surface(sc, n, __float_as_uint(a.y), __float_as_int(a.z), pix, it, k0, k1, mb, o, d, T, L, b);
}

// The same for a hit on triangle k of the scene (primitive n_prims + k): the hit point o + d * t, the
// triangle's own unit normal (computed by the host when the scene is uploaded), its material and light.
PT_DEV bool shade_bounce_triangle(const SceneView &sc, const float4 *tri4, float t, uint32_t pix, uint32_t it,
                                  uint32_t k0, uint32_t k1, int mb, V3 &o, V3 &d, V3 &T, V3 &L, int &b)
{
    const float4 t0 = tri4[0], t1 = tri4[1], t3 = tri4[3];
    o = o + d * t;
    return shade_surface(sc, mk(t3.x, t3.y, t3.z), __float_as_uint(t0.w), __float_as_int(t1.w), pix, it, k0, k1, mb, o, d, T, L, b);
}

// RenderKernel's accumulation (pathtracer.cu:101-109): c is the in-order sum of the pixel's four
// path radiances; acc the pixel's frame-buffer value.
PT_DEV V3 accumulate_pixel(V3 acc, V3 c_sum, uint32_t it, int accum_mode, V3 *c_out = nullptr)
{
    const V3 c = c_sum * 0.25f;   // c /= 4.f is c * (1.f / 4.f), exactly 0.25
    if (c_out) *c_out = c;
    if (accum_mode == 0) {
        const float inv_it = pt_rcp((float)it);
        const float w = (float)(it - 1);
        return mk((acc.x * w + c.x) * inv_it, (acc.y * w + c.y) * inv_it, (acc.z * w + c.z) * inv_it);
    }
    return acc + c;
}

} // namespace ptmi
