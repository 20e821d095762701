// Direct-light sampling for the opt-in NEE + MIS integrator: device versions of the functions
// the reference carries but never calls from Trace -- SampleLight (scene.h:150-170), Pdf
// (scene.h:110-144), Primitive::Sample (primitive.h:50-72), VisibilityTester (light.h:23-32),
// UniformSampleSphere / UniformSampleCone / UniformConePdf / PowerHeuristic
// (montecarlo.h:135-159), CoordinateSystem (geometry.h:548-558), Material::F (material.h:37-43).
//
// Three of them cannot work as written and are repaired here (the as-written text is pinned
// separately, on the CPU, against the reference's own host build):
//   - Primitive::Sample(p, ...) primitive.h:67 assigns the comparison `Intersect(r) > 0.f` to
//     thit instead of the distance; repaired as pbrt's Sphere::Sample, which it transcribes;
//   - SampleLight scene.h:167 passes -wi to Light::L, which then tests the back face and returns
//     black for every point outside the light; repaired: L(sp, wi, n);
//   - Pdf(scene, prim, p, wi) scene.h:114-124 (p on or inside the light) uses the bool of the
//     scene-wide Intersect as a distance and the normal at p; repaired as pbrt's Shape::Pdf.
// Expression order follows the reference; sqrt, 1/x and a/b are the correctly rounded device
// sequences, sin/cos the deterministic pair, so results equal the host's bit for bit.
#pragma once

#include "pt_trace.hpp"

namespace ptmi {

constexpr uint32_t STREAM_LIGHT = 2u;   // Philox stream: slot = bounce -> (u, v, light choice, -)

// UniformSampleSphere, montecarlo.h:135-142
PT_DEV V3 uniform_sample_sphere(float u1, float u2)
{
    const float z = 1.f - 2.f * u1;
    const float e = 1.f - z * z;
    const float r = pt_sqrt(0.f > e ? 0.f : e);
    const float phi = 2.f * PT_M_PI * u2;
    float s, c;
    pt_sincosf(phi, s, c);
    return mk(r * c, r * s, z);
}

// UniformSampleCone, montecarlo.h:144-150 (Lerp: globals.h:103-105)
PT_DEV V3 uniform_sample_cone(float u1, float u2, float costhetamax, V3 x, V3 y, V3 z)
{
    const float costheta = (1.f - u1) * costhetamax + u1 * 1.f;
    const float sintheta = pt_sqrt(1.f - costheta * costheta);
    const float phi = u2 * 2.f * PT_M_PI;
    float s, c;
    pt_sincosf(phi, s, c);
    return ((x * c) * sintheta + (y * s) * sintheta) + z * costheta;
}

// UniformConePdf, montecarlo.h:152-154
PT_DEV float uniform_cone_pdf(float cos_theta_max) { return pt_rcp(2.f * PT_M_PI * (1.f - cos_theta_max)); }

// PowerHeuristic(1, fPdf, 1, gPdf), montecarlo.h:156-159
PT_DEV float power_heuristic(float f_pdf, float g_pdf)
{
    const float f = 1.f * f_pdf, g = 1.f * g_pdf;
    return pt_div(f * f, f * f + g * g);
}

// CoordinateSystem, geometry.h:548-558
PT_DEV void coordinate_system(V3 v1, V3 &v2, V3 &v3)
{
    if (__builtin_fabsf(v1.x) > __builtin_fabsf(v1.y)) {
        const float inv_len = pt_rcp(pt_sqrt(v1.x * v1.x + v1.z * v1.z));
        v2 = mk(-v1.z * inv_len, 0.f, v1.x * inv_len);
    } else {
        const float inv_len = pt_rcp(pt_sqrt(v1.y * v1.y + v1.z * v1.z));
        v2 = mk(0.f, v1.z * inv_len, -v1.y * inv_len);
    }
    v3 = cross(v1, v2);
}

// Primitive::Intersect with the default bounds, primitive.h:39-45; s = (cx, cy, cz, r*r)
PT_DEV float sphere_intersect_one(float4 s, V3 o, V3 d)
{
    const V3 op = mk(s.x, s.y, s.z) - o;
    const float b = dot(op, d);
    float det = b * b - dot(op, op) + s.w;
    if (det < 0.f) return -1.f;
    det = pt_sqrt(det);
    float t;
    return (t = b - det) > PT_EPSILON ? (t < PT_INF ? t : -1.f)
                                      : ((t = b + det) > PT_EPSILON ? (t < PT_INF ? t : -1.f) : -1.f);
}

// Primitive::Sample(sample, ns), primitive.h:50-54; a = (1/r, -, -, r)
PT_DEV V3 sphere_sample_area(float4 s, float4 a, float u, float v, V3 &ns)
{
    const V3 pos = mk(s.x, s.y, s.z);
    const V3 p = pos + uniform_sample_sphere(u, v) * a.w;
    ns = (p - pos) * a.x;
    return p;
}

// Primitive::Sample(p, sample, ns), primitive.h:55-72, line 67 repaired
PT_DEV V3 sphere_sample_from(float4 s, float4 a, V3 p, float u, float v, V3 &ns)
{
    const V3 pos = mk(s.x, s.y, s.z);
    const V3 wc = normalize(pos - p);
    V3 wcx, wcy;
    coordinate_system(wc, wcx, wcy);
    const V3 dp = p - pos;
    const float dist2 = dot(dp, dp);
    if (dist2 - s.w < 1e-4f) return sphere_sample_area(s, a, u, v, ns);
    const float sin_theta_max2 = pt_div(s.w, dist2);
    const float e = 1.f - sin_theta_max2;
    const float cos_theta_max = pt_sqrt(0.f > e ? 0.f : e);
    const V3 d = uniform_sample_cone(u, v, cos_theta_max, wcx, wcy, wc);
    float thit = sphere_intersect_one(s, p, d);
    if (!(thit > 0.f)) thit = dot(pos - p, normalize(d));
    const V3 ps = p + d * thit;
    ns = normalize(ps - pos);
    return ps;
}

// Pdf(scene, prim, p, wi), scene.h:110-130, the inside branch repaired
PT_DEV float sphere_light_pdf(float4 s, float4 a, V3 p, V3 wi)
{
    const V3 pos = mk(s.x, s.y, s.z);
    const V3 dp = p - pos;
    const float dist2 = dot(dp, dp);
    if (dist2 - s.w < 1e-4f) {
        const float thit = sphere_intersect_one(s, p, wi);
        if (thit < 0.f) return 0.f;
        const V3 ph = p + wi * thit;
        const V3 n = (ph - pos) * a.x;
        const V3 dq = p - ph;
        const float area = 4 * PT_M_PI * a.w * a.w;                          // Primitive::Area, primitive.h:73
        float pdf = pt_div(dot(dq, dq), __builtin_fabsf(dot(n, -wi)) * area);
        if (__builtin_isinf(pdf)) pdf = 0.f;
        return pdf;
    }
    const float sin_theta_max2 = pt_div(s.w, dist2);
    const float e = 1.f - sin_theta_max2;
    const float cos_theta_max = pt_sqrt(0.f > e ? 0.f : e);
    return uniform_cone_pdf(cos_theta_max);
}

struct Segment { V3 d; float maxt; };

// VisibilityTester::SetSegment, light.h:24-29 (origin p1 and mint = eps1 are the caller's)
PT_DEV Segment visibility_segment(V3 p1, V3 p2, float eps2)
{
    Segment s;
    const float dist = length(p1 - p2);
    s.maxt = dist * (1.f - eps2);
    s.d = vdiv(p2 - p1, dist);
    return s;
}

// Per-lane state of the NEE integrator between the passes of the render loop.
struct NeeState {
    bool shadow;          // the lane's next query is the shadow ray of the vertex just shaded
    bool end_after;       // ... and the path ends once it is answered
    bool specular_bounce; // pathtracer.cu:117,148
    float bsdf_pdf_prev;
    V3 sh_d;              // shadow ray direction (origin = o)
    float sh_tmax;        // its upper bound (point lights), PT_INF for area lights
    int sh_prim;          // area light: the light's sphere, which the nearest hit must be; -1: point light
    V3 pending;           // throughput x direct light, added if the light is visible
};

// One iteration of Trace's loop (pathtracer.cu:125-168) with next-event estimation and multiple
// importance sampling, the integrator the commented-out lines pathtracer.cu:117,133,148 prepare,
// built like pbrt's PathIntegrator with one BSDF sample serving both the continuation and the
// BSDF side of the MIS pair:
//   - emission found by a camera ray or after a specular / transmissive bounce counts in full;
//     after a diffuse bounce it is weighted by PowerHeuristic(bsdfPdf, lightPdf / nLights);
//   - at a diffuse vertex one light is chosen uniformly and sampled, weighted by
//     PowerHeuristic(lightPdf / nLights, bsdfPdf), provided the material's Pdf for that direction
//     is positive (directions the BSDF sampler rejects count nothing, scene.h:139,184, so that
//     the expectation equals Trace's).  The shadow query is left to the caller's next Intersect:
//     an area light is visible when the nearest hit along the direction is its own sphere -- the
//     rule by which a BSDF-sampled ray finds emission; VisibilityTester's segment (light.h:24-29)
//     ends EPSILON = 3 % short of the light and overlooks the demo scene's ceiling, 0.4 units in
//     front of the light sphere -- a point light when nothing lies within the segment.
// Returns true when the path ends (after the pending shadow query, if st.shadow is set).
PT_DEV bool shade_bounce_nee(const SceneParams &sp, const SceneView &sc, float4 s, int best, float t, uint32_t pix,
                             uint32_t it, uint32_t k0, uint32_t k1, int mb, V3 &o, V3 &d, V3 &T, V3 &L, int &b,
                             NeeState &st)
{
    const float4 a = sc.aux[best];
    const V3 P = o + d * t;
    const V3 n = (P - mk(s.x, s.y, s.z)) * a.x;
    const V3 wo = d;
    const uint32_t mat_id = __float_as_uint(a.y);
    const int light_id = __float_as_int(a.z);
    const float inv_nl = sp.n_lights ? pt_rcp((float)sp.n_lights) : 0.f;

    if (light_id > -1) {
        const float4 li = sc.light[light_id];
        const V3 le = dot(n, -wo) > 0.f ? mk(li.x, li.y, li.z) : mk(0.f, 0.f, 0.f);
        if (b == 0 || st.specular_bounce) {
            L = L + T * le;
        } else {
            // o is still the previous vertex, wo the direction sampled there
            const float lp = sphere_light_pdf(s, a, o, wo) * inv_nl;
            const float w = power_heuristic(st.bsdf_pdf_prev, lp);
            L = L + T * (le * w);
        }
    }

    const float4 m = sc.mat[mat_id];
    const int2 mt = sc.mat_type[mat_id];
    const int mtype = mt.x;

    st.shadow = false;
    if (mtype == 0 && sp.n_lights > 0) {
        uint32_t rl[4];
        philox4x32_10(pix, it, (uint32_t)b, STREAM_LIGHT, k0, k1, rl);
        uint32_t j = (uint32_t)(u01(rl[2]) * (float)sp.n_lights);
        if (j > sp.n_lights - 1) j = sp.n_lights - 1;
        const float4 lg = sp.light_geom[j];
        const float4 lc = sp.light[j];
        V3 wl, li;
        float pdf_l;
        Segment seg;
        const int ltype = __float_as_int(lg.w);
        const bool point = ltype == 0;
        if (ltype != 0 && ltype != 1) {
            // SampleLight, scene.h:169: a light of unknown type contributes nothing
            wl = mk(0.f, 0.f, 1.f); li = mk(0.f, 0.f, 0.f); pdf_l = 0.f;
            seg.d = wl; seg.maxt = 0.f;
        } else if (point) {
            // SampleLight, scene.h:153-158
            const V3 pos = mk(lg.x, lg.y, lg.z);
            wl = normalize(pos - P);
            pdf_l = 1.f;
            seg = visibility_segment(P, pos, 0.f);
            const V3 dp = pos - P;
            li = vdiv(mk(lc.x, lc.y, lc.z), dot(dp, dp));
            st.sh_prim = -1;
            st.sh_tmax = seg.maxt;
        } else {
            // SampleLight, scene.h:160-168
            const uint32_t prim = __float_as_uint(lg.x);
            const float4 ls = sp.sph[prim], la = sp.aux[prim];
            V3 ln;
            const V3 spnt = sphere_sample_from(ls, la, P, u01(rl[0]), u01(rl[1]), ln);
            wl = normalize(spnt - P);
            pdf_l = sphere_light_pdf(ls, la, P, wl);
            seg = visibility_segment(P, spnt, PT_EPSILON);
            li = dot(ln, -wl) > 0.f ? mk(lc.x, lc.y, lc.z) : mk(0.f, 0.f, 0.f);
            st.sh_prim = (int)prim;
            st.sh_tmax = PT_INF;
        }
        if (pdf_l > 0.f && !(li.x == 0.f && li.y == 0.f && li.z == 0.f)) {
            const V3 fl = mk(m.x, m.y, m.z);   // Material::F, material.h:37-43: the table holds 0 + color/pi for diffuse materials
            const float bp = -dot(wo, wl) > 0.f ? dot(wl, n) * PT_INV_PI : 0.f;   // Pdf(mat, ...), scene.h:136-144
            if (!(fl.x == 0.f && fl.y == 0.f && fl.z == 0.f) && bp > 0.f) {
                const float pl = pdf_l * inv_nl;
                const float w = point ? 1.f : power_heuristic(pl, bp);
                const V3 ld = (fl * li) * pt_div(__builtin_fabsf(dot(wl, n)) * w, pl);
                st.pending = T * ld;
                st.sh_d = seg.d;
                st.shadow = true;
            }
        }
    }

    // the path's own random numbers only now: they hold no registers during the light sampling
    uint32_t r[4];
    philox4x32_10(pix, it, (uint32_t)b, STREAM_PATH, k0, k1, r);
    V3 wi;
    float pdf;
    const V3 f = sample_material(mtype, mk(m.x, m.y, m.z), m.w, __int_as_float(mt.y), wo, wi, pdf, n, u01(r[0]), u01(r[1]));
    o = P;   // the shadow ray starts here as well
    if ((f.x == 0.f && f.y == 0.f && f.z == 0.f) || pdf == 0.f) return true;
    st.specular_bounce = mtype != 0;
    st.bsdf_pdf_prev = pdf;
    T = T * vdiv(f * __builtin_fabsf(dot(wi, n)), pdf);
    d = wi;
    bool end = false;
    if (b > 3) {
        const float gb = T.y > T.z ? T.y : T.z;
        const float mx = T.x > gb ? T.x : gb;
        const float q = .5f < mx ? .5f : mx;
        if (u01(r[2]) > q) end = true;
        else T = vdiv(T, q);
    }
    if (b == mb) end = true;
    ++b;
    return end;
}

} // namespace ptmi
