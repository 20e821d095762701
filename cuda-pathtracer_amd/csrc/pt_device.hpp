// Device-side building blocks of the trace loop for gfx950 (wave64).
//
// Everything here is fp32 evaluated in the reference's expression order with
// -ffp-contract=off, IEEE division and square root: the demo scene's wall spheres have
// radius 1e5, so `b*b - op.op + r*r` (primitive.h:42) cancels catastrophically and one fused
// or reordered operation changes which sphere is nearest.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptmi {

// globals.h:44-49,56-60
#define PT_M_PI    3.14159265358979323846f
#define PT_INV_PI  0.31830988618379067154f
#define PT_EPSILON 3E-2f
#define PT_INF     3.402823466e+38f

#define PT_DEV __device__ __forceinline__

// Measurement builds (-DPT_LANE_STATS, tools/stats): PT_LS(i) counts, per code site i, how often a
// wave executes the site and with how many active lanes.  Empty in the product build.
#ifdef PT_LANE_STATS
static __device__ unsigned long long g_lane_stats[2 * 32];
#define PT_LS(i)                                                                                       \
    do {                                                                                               \
        const unsigned long long m_ = __ballot(1);                                                     \
        if ((int)__lane_id() == __ffsll(m_) - 1) {                                                     \
            atomicAdd(&g_lane_stats[2 * (i)], 1ull);                                                   \
            atomicAdd(&g_lane_stats[2 * (i) + 1], (unsigned long long)__popcll(m_));                   \
        }                                                                                              \
    } while (0)
#else
#define PT_LS(i) ((void)0)
#endif

// ---- correctly rounded sqrt, 1/x and a/b ------------------------------------------------------
// hipcc's IEEE expansions of sqrtf and '/' carry denormal pre-scaling, class checks and
// div_scale/div_fixup steps made of half-rate compare/select instructions (~55 and ~40 issue
// cycles).  For operands in a comfortable exponent range the Newton/Markstein sequences below
// give the same correctly rounded result in ~22 / ~17 / ~24 cycles of full-rate FMAs; outside
// the range (zero, denormal, huge, inf, NaN, negative sqrt argument) the generic expansion
// runs.  Correct rounding makes them bit-identical to the host's sqrtf and '/'; pt_selftest
// checks every float (sqrt, 1/x) and 2^32 random pairs (a/b) against the generic expansion.
PT_DEV bool in_exp_range(float x, uint32_t lo_biased, uint32_t hi_biased)
{
    // true iff lo <= biased exponent of |x| < hi (sign ignored)
    return ((__float_as_uint(x) & 0x7fffffffu) - (lo_biased << 23)) < ((hi_biased - lo_biased) << 23);
}

PT_DEV float sqrt_newton(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    h = __builtin_fmaf(h, e, h);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

PT_DEV float pt_sqrt(float x)
{
    // positive, 2^-100 <= x < 2^100 (a set sign bit makes the unsigned difference huge)
    if (__builtin_expect((__float_as_uint(x) - (27u << 23)) < (200u << 23), 1)) return sqrt_newton(x);
    return __builtin_sqrtf(x);
}

// pt_sqrt for the discriminant of a sphere test, where the root only feeds t = b -+ root and the
// test eps < t < tmax with a finite tmax (primitive.h:43-44, scene.h:78-80).  The short sequence is
// correctly rounded for every finite x >= 2^-100 (measured: it loses bits below 2^-102, where its
// residual turns denormal, and nowhere above), so ONE comparison gates it; what the comparison
// lets through besides is +infinity, for which the sequence returns NaN instead of +infinity -- and
// both make t infinite or NaN, which fails the test either way.  (NaN and everything below 2^-100,
// zero and the negative numbers included, compare false and take the generic expansion.)
PT_DEV float pt_sqrt_discriminant(float x)
{
    if (__builtin_expect(x > 7.8886090522101181e-31f, 1)) return sqrt_newton(x);   // 2^-100
    return __builtin_sqrtf(x);
}

PT_DEV float rcp_newton(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.f);
    r = __builtin_fmaf(e, r, r);
    e = __builtin_fmaf(-x, r, 1.f);
    return __builtin_fmaf(e, r, r);
}

PT_DEV float pt_rcp(float x)
{
    if (__builtin_expect(in_exp_range(x, 27u, 227u), 1)) return rcp_newton(x);
    return 1.f / x;
}

PT_DEV float pt_div(float a, float b)
{
    // 2^-30 <= |a|, |b| < 2^30: quotient and the exact residual a - b*q stay representable
    if (__builtin_expect(in_exp_range(a, 97u, 157u) && in_exp_range(b, 97u, 157u), 1)) {
        const float r = rcp_newton(b);
        const float q = a * r;
        const float e = __builtin_fmaf(-b, q, a);
        return __builtin_fmaf(e, r, q);
    }
    return a / b;
}

// pt_div for operands known to lie in its fast range: 2^-30 <= |b| < 2^30 and a == 0 or in the
// same range (no test, no fallback; b == 0 gives NaN).
PT_DEV float pt_div_in_range(float a, float b)
{
    const float r = rcp_newton(b);
    const float q = a * r;
    const float e = __builtin_fmaf(-b, q, a);
    return __builtin_fmaf(e, r, q);
}

struct V3 { float x, y, z; };

// geometry.h:91-133: Vec / Point / Color arithmetic; `a / s` is `a * (1.f / s)`.
PT_DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV V3 operator*(V3 a, float d) { return mk(a.x * d, a.y * d, a.z * d); }
PT_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
PT_DEV V3 vdiv(V3 a, float d) { const float inv = pt_rcp(d); return a * inv; }
PT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      // geometry.h:48-50
PT_DEV V3 cross(V3 a, V3 b)                                                      // geometry.h:45-47
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PT_DEV float length(V3 a) { return pt_sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
// geometry.h:131-133: a * (1.f / Length()).  One range test serves the square root and the
// reciprocal: for 2^-100 <= |a|^2 < 2^100 the length lies in [2^-50, 2^50), inside the reciprocal's
// fast range; outside, both take their generic expansions -- the same correctly rounded values as
// pt_rcp(pt_sqrt(.)) returns, whichever of their paths those would have taken.
PT_DEV V3 normalize(V3 a)
{
    const float len2 = a.x * a.x + a.y * a.y + a.z * a.z;
    float inv;
    if (__builtin_expect((__float_as_uint(len2) - (27u << 23)) < (200u << 23), 1)) {
        inv = rcp_newton(sqrt_newton(len2));
    } else {
        inv = 1.f / __builtin_sqrtf(len2);
    }
    return a * inv;
}

// ---- counter-based RNG --------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  ctr = (pixel, iteration, slot, stream), key = seed.
PT_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // a ^ b ^ c in one v_bitop3_b32 (truth table 0x96) instead of two v_xor_b32
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

PT_DEV float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }

enum : uint32_t { STREAM_CAMERA = 0u, STREAM_PATH = 1u };

// ---- deterministic sine/cosine ------------------------------------------------------------
// Replaces sinf/cosf of ConcentricSampleDisk (montecarlo.h:116-117) for theta in [0, 2*pi]:
// a fixed sequence of fp32 multiplies and adds that the host evaluates bit-identically
// (three-term Cody-Waite reduction by pi/2, Cephes minimax kernels on [-pi/4, pi/4]).
PT_DEV void pt_sincosf(float x, float &s_out, float &c_out)
{
    const float two_over_pi = 0.636619772367581343f;
    const float P1 = 1.5703125f;
    const float P2 = 4.837512969970703125e-4f;
    const float P3 = 7.54978995489188216e-8f;
    const int k = (int)(x * two_over_pi + 0.5f);
    const float kf = (float)k;
    const float r = ((x - kf * P1) - kf * P2) - kf * P3;
    const float z = r * r;
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z;
    const float s = r + r * ps;
    const float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f);
    const float c = (1.0f - 0.5f * z) + (z * z) * pc;
    const bool swap = (k & 1) != 0;
    const float sa = swap ? c : s;
    const float ca = swap ? s : c;
    s_out = (k & 2) ? -sa : sa;
    c_out = ((k + 1) & 2) ? -ca : ca;
}

// ---- sampling -----------------------------------------------------------------------------
// ConcentricSampleDisk, montecarlo.h:76-118.  The four regions are folded into selects and ONE
// division: region 1 gives sy/r or 8 + sy/r, region 2: 2 - sx/r, region 3: 4 - sy/r, region 4:
// 6 + sx/r; `0 + q` equals q for the positive quotient of region 1, and a - q == a + (-q).
PT_DEV void concentric_sample_disk(float u1, float u2, float &dx, float &dy)
{
    const float sx = 2.f * u1 - 1.f;
    const float sy = 2.f * u2 - 1.f;
    const bool upper = sx >= -sy;            // regions 1, 2
    const bool r1 = upper && sx > sy;
    const bool r2 = upper && !r1;
    const bool r3 = !upper && sx <= sy;
    const bool use_sy = r1 || r3;            // numerator sy (else sx)
    const float num = use_sy ? sy : sx;
    const float r = r1 ? sx : (r2 ? sy : (r3 ? -sx : -sy));
    const float off = r1 ? (sy > 0.f ? 0.f : 8.f) : (r2 ? 2.f : (r3 ? 4.f : 6.f));
    const float q = pt_div(num, r);
    float theta = (r2 || r3) ? off - q : off + q;
    theta *= PT_M_PI / 4.f;
    float s, c;
    pt_sincosf(theta, s, c);
    const bool origin = sx == 0.f && sy == 0.f;   // montecarlo.h:85-89
    dx = origin ? 0.f : r * c;
    dy = origin ? 0.f : r * s;
}

// RotateByNormal, montecarlo.h:120-125 (tangent u deliberately not normalised)
PT_DEV V3 rotate_by_normal(V3 wo, V3 n)
{
    const V3 w = n;
    const V3 u = __builtin_fabsf(n.x) > __builtin_fabsf(n.z) ? mk(-n.y, n.x, 0.f) : mk(0.f, -n.z, n.y);
    const V3 v = cross(w, u);
    return normalize((u * wo.x + v * wo.y) + w * wo.z);
}

// CosineSampleHemisphere, montecarlo.h:127-133 (max is the ternary macro, globals.h:67)
PT_DEV V3 cosine_sample_hemisphere(float u1, float u2, V3 n)
{
    V3 ret = mk(0.f, 0.f, 0.f);
    concentric_sample_disk(u1, u2, ret.x, ret.y);
    const float e = 1.f - ret.x * ret.x - ret.y * ret.y;
    ret.z = pt_sqrt(0.f > e ? 0.f : e);
    return rotate_by_normal(ret, n);
}

// Reflect, globals.h:107-109
PT_DEV V3 reflect(V3 wo, V3 n) { return wo - n * (2.f * dot(wo, n)); }

// Refract, globals.h:111-115
PT_DEV V3 refract(V3 wo, V3 n, float eta)
{
    const float cost = dot(wo, n);
    const float cos2t = 1.f - eta * eta * (1.f - cost * cost);
    return normalize(wo * eta - n * (eta * cost + pt_sqrt(cos2t)));
}

// reflectance (Fresnel), globals.h:117-126
PT_DEV float reflectance(V3 inc, V3 nor, float n1, float n2)
{
    const float n = pt_div(n1, n2);
    const float cosI = -(dot(nor, inc));
    const float sinT2 = n * n * (1.0f - cosI * cosI);
    if (sinT2 > 1.0f) return 1.0f;
    const float cosT = pt_sqrt(1.0f - sinT2);
    const float rOrth = pt_div(n1 * cosI - n2 * cosT, n1 * cosI + n2 * cosT);
    const float rPar = pt_div(n2 * cosI - n1 * cosT, n2 * cosI + n1 * cosT);
    return (rOrth * rOrth + rPar * rPar) / 2.0f;
}

// SampleMaterial, scene.h:177-221 with Material::F, material.h:37-43.
//
// A wave executes every branch that any of its lanes takes, so the refraction branch -- a few
// lanes per wave, but almost always some -- used to cost a fifth of the demo render.  The three
// BSDFs are therefore evaluated in common stages: the expensive operations that the diffuse
// branch needs anyway (one division, one square root, one normalisation) also serve the glass
// lanes, and only what is left (two Fresnel quotients) runs for the glass lanes alone.  Every
// value is produced by the same IEEE operations on the same operands as in the reference's
// order; the identities used are exact in fp32:
//   - n1/n2 is 1/coef from outside and coef/1 == coef from inside (scene.h:199-209); 1/coef depends
//     on the material alone and comes from the material table (inv_coef, computed by the host);
//   - x*1.f == x, so with n1 or n2 equal to 1 the two Fresnel quotients (globals.h:122-123) are
//     X = (cosI - coef*cosT)/(cosI + coef*cosT) and Y = (coef*cosI - cosT)/(coef*cosI + cosT),
//     (rOrth, rPar) = (X, Y) from outside and (Y, X) from inside, and X*X + Y*Y == Y*Y + X*X;
//   - Refract's cost = wo.n is -cosI, so its cos2t = 1 - eta*eta*(1 - cost*cost) is the very
//     value 1 - sinT2 whose root reflectance() already took: sqrt(cos2t) == cosT.
PT_DEV V3 sample_material(int type, V3 f_mat, float coef, float inv_coef, V3 wo, V3 &wi, float &pdf, V3 n,
                          float su, float sv)
{
    const bool is_d = type == 0, is_s = type == 1, is_g = type == 2;

    // ---- glass: side of the surface, scene.h:198-209 (nnor == n for the other materials)
    const bool outside = dot(wo, n) < 0.f;
    const bool flip = is_g && !outside;
    const V3 nnor = mk(flip ? -n.x : n.x, flip ? -n.y : n.y, flip ? -n.z : n.z);

    // ---- diffuse: ConcentricSampleDisk up to its division, montecarlo.h:76-114
    const float sx = 2.f * su - 1.f;
    const float sy = 2.f * sv - 1.f;
    // Regions 1..4 of montecarlo.h:91-113: 1 = upper && gt, 2 = upper && !gt, 3 = !upper && !gt,
    // 4 = !upper && gt (sx and sy are finite, so `sx <= sy` is !(sx > sy)).  The radius is sx, sy,
    // -sx, -sy there: the component of the larger magnitude, which is positive in regions 1, 2
    // and negative in 3, 4 -- its absolute value in all four.
    const bool upper = sx >= -sy;
    const bool gt = sx > sy;
    const bool xdom = upper == gt;           // regions 1, 3
    const float num = xdom ? sy : sx;
    const float r = __builtin_fabsf(xdom ? sx : sy);
    const float off = xdom ? (upper ? (sy > 0.f ? 0.f : 8.f) : 4.f) : (upper ? 2.f : 6.f);

    // ---- division 1 (diffuse): num/r.  Both are multiples of 2^-23 with |num| <= r <= 1; r == 0 only
    // at the disk's origin, where the quotient is not used.
    const float q = pt_div_in_range(num, r);

    float root_arg, cosI = 0.f, eta = 1.f;
    bool tir = false;
    V3 ret = mk(0.f, 0.f, 0.f);
    PT_LS(10);
    if (is_g) {
        PT_LS(11);
        // reflectance(), globals.h:117-121
        eta = outside ? inv_coef : coef;
        cosI = -(dot(nnor, wo));
        const float sinT2 = eta * eta * (1.0f - cosI * cosI);
        tir = sinT2 > 1.0f;
        root_arg = tir ? 1.f : 1.0f - sinT2;   // total internal reflection: the root is not used
    } else {
        PT_LS(12);
        // montecarlo.h:115-117, 129-130
        float theta = gt ? off + q : off - q;   // minus in regions 2, 3
        theta *= PT_M_PI / 4.f;
        // montecarlo.h:85-89: (0, 0) maps to (0, 0).  There r == 0 and q is 0/0; with theta = 0 the
        // sine and cosine are exactly 0 and 1, and r * c, r * s are the +0 the reference returns.
        if (r == 0.f) theta = 0.f;
        float s, c;
        pt_sincosf(theta, s, c);
        ret.x = r * c;
        ret.y = r * s;
        const float e = 1.f - ret.x * ret.x - ret.y * ret.y;
        root_arg = 0.f > e ? 0.f : e;
    }

    // ---- square root: diffuse z of the hemisphere sample | glass cosT
    const float root = pt_sqrt(root_arg);

    V3 vec;              // direction before normalisation
    bool do_reflect = is_s;
    if (is_g) {
        PT_LS(13);
        // reflectance(), globals.h:122-125
        const float cosT = root;
        const float x = pt_div(cosI - coef * cosT, cosI + coef * cosT);
        const float y = pt_div(coef * cosI - cosT, coef * cosI + cosT);
        const float rr = outside ? (x * x + y * y) : (y * y + x * x);
        const float refl = tir ? 1.0f : rr / 2.0f;
        do_reflect = su < refl;
        // Refract, globals.h:111-115, before its Normalize(): cost == -cosI, sqrt(cos2t) == cosT
        vec = wo * eta - nnor * (eta * (-cosI) + cosT);
    } else {
        PT_LS(14);
        // RotateByNormal, montecarlo.h:120-124, before its Normalize()
        ret.z = root;
        const V3 w = n;
        const V3 u = __builtin_fabsf(n.x) > __builtin_fabsf(n.z) ? mk(-n.y, n.x, 0.f) : mk(0.f, -n.z, n.y);
        const V3 v = cross(w, u);
        vec = (u * ret.x + v * ret.y) + w * ret.z;
    }

    // ---- normalisation (diffuse, refraction) or mirror direction (specular, Fresnel reflection)
    if (do_reflect) { PT_LS(15); wi = reflect(wo, nnor); }
    else            { PT_LS(16); wi = normalize(vec); }

    // pdf: scene.h:185, 190, 217.  f (scene.h:186 with Material::F, material.h:37-43: 0 + color/pi;
    // 191: coef * color; 216: color) depends on the material alone: the host computes it with the
    // same fp32 operations when the scene is uploaded and f_mat is that value (black for an unknown
    // material type, which ends the path, pathtracer.cu:146, before wi or pdf are looked at).
    pdf = is_d ? (dot(wo, wi) < 0.f ? __builtin_fabsf(dot(wi, n)) * PT_INV_PI : 0.f) : 1.f;
    return f_mat;
}

// ---- nearest hit --------------------------------------------------------------------------
// One sphere of the scan in Intersect (scene.h:75-91) with Primitive::Intersect
// (primitive.h:39-45) inlined.  s = (cx, cy, cz, radius*radius).  The strict `t < tmax`
// keeps the lowest index on equal distance when spheres are visited in index order.
PT_DEV void sphere_test(float4 s, int index, V3 o, V3 d, float &tmax, int &best)
{
    const float opx = s.x - o.x, opy = s.y - o.y, opz = s.z - o.z;
    const float b = opx * d.x + opy * d.y + opz * d.z;
    const float det = b * b - (opx * opx + opy * opy + opz * opz) + s.w;
    PT_LS(2);
    if (!(det < 0.f)) {
        PT_LS(3);
        const float sq = pt_sqrt_discriminant(det);
        float t = b - sq;
        if (!(t > PT_EPSILON)) t = b + sq;
        if (t > PT_EPSILON && t < tmax) { tmax = t; best = index; }
    }
}

} // namespace ptmi
