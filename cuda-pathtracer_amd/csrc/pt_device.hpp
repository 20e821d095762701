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

struct V3 { float x, y, z; };

// geometry.h:91-133: Vec / Point / Color arithmetic; `a / s` is `a * (1.f / s)`.
PT_DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV V3 operator*(V3 a, float d) { return mk(a.x * d, a.y * d, a.z * d); }
PT_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
PT_DEV V3 vdiv(V3 a, float d) { const float inv = 1.f / d; return a * inv; }
PT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }      // geometry.h:48-50
PT_DEV V3 cross(V3 a, V3 b)                                                      // geometry.h:45-47
{
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PT_DEV float length(V3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
PT_DEV V3 normalize(V3 a) { return vdiv(a, length(a)); }                         // geometry.h:131-133

// ---- counter-based RNG --------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  ctr = (pixel, iteration, slot, stream), key = seed.
PT_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
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
// ConcentricSampleDisk, montecarlo.h:76-118
PT_DEV void concentric_sample_disk(float u1, float u2, float &dx, float &dy)
{
    float r, theta;
    const float sx = 2.f * u1 - 1.f;
    const float sy = 2.f * u2 - 1.f;
    if (sx == 0.f && sy == 0.f) { dx = 0.f; dy = 0.f; return; }
    if (sx >= -sy) {
        if (sx > sy) {
            r = sx;
            if (sy > 0.f) theta = sy / r;
            else          theta = 8.f + sy / r;
        } else {
            r = sy;
            theta = 2.f - sx / r;
        }
    } else {
        if (sx <= sy) {
            r = -sx;
            theta = 4.f - sy / r;
        } else {
            r = -sy;
            theta = 6.f + sx / r;
        }
    }
    theta *= PT_M_PI / 4.f;
    float s, c;
    pt_sincosf(theta, s, c);
    dx = r * c;
    dy = r * s;
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
    ret.z = __builtin_sqrtf(0.f > e ? 0.f : e);
    return rotate_by_normal(ret, n);
}

// Reflect, globals.h:107-109
PT_DEV V3 reflect(V3 wo, V3 n) { return wo - n * (2.f * dot(wo, n)); }

// Refract, globals.h:111-115
PT_DEV V3 refract(V3 wo, V3 n, float eta)
{
    const float cost = dot(wo, n);
    const float cos2t = 1.f - eta * eta * (1.f - cost * cost);
    return normalize(wo * eta - n * (eta * cost + __builtin_sqrtf(cos2t)));
}

// reflectance (Fresnel), globals.h:117-126
PT_DEV float reflectance(V3 inc, V3 nor, float n1, float n2)
{
    const float n = n1 / n2;
    const float cosI = -(dot(nor, inc));
    const float sinT2 = n * n * (1.0f - cosI * cosI);
    if (sinT2 > 1.0f) return 1.0f;
    const float cosT = __builtin_sqrtf(1.0f - sinT2);
    const float rOrth = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
    const float rPar = (n2 * cosI - n1 * cosT) / (n2 * cosI + n1 * cosT);
    return (rOrth * rOrth + rPar * rPar) / 2.0f;
}

// SampleMaterial, scene.h:177-221 with Material::F, material.h:37-43
PT_DEV V3 sample_material(int type, V3 color, float coef, V3 wo, V3 &wi, float &pdf, V3 n,
                          float su, float sv)
{
    V3 f = mk(0.f, 0.f, 0.f);
    if (type == 0) {
        wi = cosine_sample_hemisphere(su, sv, n);
        pdf = dot(wo, wi) < 0.f ? __builtin_fabsf(dot(wi, n)) * PT_INV_PI : 0.f;
        f = f + color * PT_INV_PI;
    } else if (type == 1) {
        wi = reflect(wo, n);
        pdf = 1.f;
        f = color * coef;
    } else if (type == 2) {
        float n1, n2;
        V3 nnor;
        if (dot(wo, n) < 0.f) { n1 = 1.f; n2 = coef; nnor = n; }
        else                  { n1 = coef; n2 = 1.f; nnor = -n; }
        const float refl = reflectance(wo, nnor, n1, n2);
        if (su < refl) wi = reflect(wo, nnor);
        else           wi = refract(wo, nnor, n1 / n2);
        f = color;
        pdf = 1.f;
    } else {
        wi = mk(0.f, 0.f, 0.f);
        pdf = 0.f;
    }
    return f;
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
    if (!(det < 0.f)) {
        const float sq = __builtin_sqrtf(det);
        float t = b - sq;
        if (!(t > PT_EPSILON)) t = b + sq;
        if (t > PT_EPSILON && t < tmax) { tmax = t; best = index; }
    }
}

} // namespace ptmi
