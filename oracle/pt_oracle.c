/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle (see pt_oracle.h).  Plain C11, scalar fp32, no
 * fused multiply-add: build with -ffp-contract=off.  Every function cites the reference
 * lines it restates; expression trees are kept in the reference's evaluation order because
 * the wall spheres of the demo scene (radius 1e5) cancel catastrophically in fp32 and a
 * single differently rounded operation flips nearest-hit decisions.
 */
#include "pt_oracle.h"
#include "pt_oracle_math.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* globals.h:44-49,56-60 */
#define PT_M_PI    3.14159265358979323846f
#define PT_INV_PI  0.31830988618379067154f
#define PT_EPSILON 3E-2f
#define PT_INF     FLT_MAX

typedef struct { float x, y, z; } v3;

/* geometry.h:91-133: Vec/Point/Color arithmetic; `a / s` is `a * (1.f / s)` (geometry.h:124-126). */
static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float d) { return v3_make(a.x * d, a.y * d, a.z * d); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline v3 v3_div(v3 a, float d) { const float inv = 1.f / d; return v3_scale(a, inv); }
/* geometry.h:48-50 */
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
/* geometry.h:45-47 */
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* geometry.h:51-53 */
static inline float v3_length(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
/* geometry.h:131-133 */
static inline v3 v3_normalize(v3 a) { return v3_div(a, v3_length(a)); }
static inline v3 v3_load(const float *p) { return v3_make(p[0], p[1], p[2]); }

typedef struct {
    const ptor_camera *cam;
    const ptor_material *mats;
    const ptor_sphere *prims;
    const ptor_light *lights;
    uint32_t n_mats, n_prims, n_lights;
    uint64_t seed;
    int max_bounces;
    int integrator;
    const ptor_triangle *tris; /* this build's extension, see pt_oracle.h; indices n_prims .. n_prims + n_tris - 1 */
    uint32_t n_tris;
} scene_ctx;

typedef struct {
    int prim; /* index, -1 when nothing was hit */
    v3 p, n;
    float t;
    uint32_t material_id; /* of the primitive hit (prim->materialId, pathtracer.cu:127) */
    int32_t light_id;
} isect_t;

/* Primitive::Intersect, primitive.h:39-45 */
static inline float sphere_intersect(const ptor_sphere *s, v3 o, v3 d, float tmin, float tmax)
{
    v3 op = v3_sub(v3_load(s->pos), o);
    float t, b = v3_dot(op, d);
    float det = b * b - v3_dot(op, op) + s->radius * s->radius;
    if (det < 0.f) return -1.f; else det = sqrtf(det);
    return (t = b - det) > tmin ? (t < tmax ? t : -1.f)
                                : ((t = b + det) > tmin ? (t < tmax ? t : -1.f) : -1.f);
}

/* A triangle in the part of Primitive::Intersect (primitive.h:39-45: t within (tmin, tmax), else -1).
 * No counterpart in the reference (parity unpinned).  Moeller-Trumbore (J. Graphics Tools 2(1), 1997)
 * in fp32, every operation in the order written, no fused multiply-add, both sides of the triangle.
 * A degenerate triangle or a ray in its plane gives det == 0, 1/det infinite and u infinite or NaN:
 * rejected by the first test like every NaN. */
static inline float triangle_intersect(const ptor_triangle *tr, v3 o, v3 d, float tmin, float tmax)
{
    const v3 v0 = v3_load(tr->v0);
    const v3 e1 = v3_sub(v3_load(tr->v1), v0), e2 = v3_sub(v3_load(tr->v2), v0);
    const v3 p = v3_cross(d, e2);
    const float det = v3_dot(e1, p);
    const float inv = 1.f / det;
    const v3 s = v3_sub(o, v0);
    const float u = v3_dot(s, p) * inv;
    if (!(u >= 0.f && u <= 1.f)) return -1.f;
    const v3 q = v3_cross(s, e1);
    const float v = v3_dot(d, q) * inv;
    if (!(v >= 0.f && u + v <= 1.f)) return -1.f;
    const float t = v3_dot(e2, q) * inv;
    return (t > tmin && t < tmax) ? t : -1.f;
}

float ptor_triangle_intersect(const ptor_triangle *tri, const float o[3], const float d[3], float tmin, float tmax)
{
    return triangle_intersect(tri, v3_load(o), v3_load(d), tmin, tmax);
}

/* Intersect, scene.h:71-94 (always called with &isect and the default tmin/tmax). */
static int scene_intersect(const scene_ctx *sc, v3 o, v3 d, isect_t *isect)
{
    float tmin = PT_EPSILON, tmax = PT_INF;
    isect_t is;
    int found = 0;
    is.prim = -1; is.p = v3_make(0, 0, 0); is.n = v3_make(0, 0, 0); is.t = 0.f;
    is.material_id = 0; is.light_id = -1;
    for (uint32_t i = 0; i < sc->n_prims; ++i) {
        float thit;
        const ptor_sphere *prim = &sc->prims[i];
        if ((thit = sphere_intersect(prim, o, d, tmin, tmax)) > 0.f) {
            tmax = thit;
            found = 1;
            is.prim = (int)i;
            is.p = v3_add(o, v3_scale(d, thit));                    /* geometry.h:343 */
            is.n = v3_div(v3_sub(is.p, v3_load(prim->pos)), prim->radius); /* primitive.h:74 */
            is.t = thit;
            is.material_id = prim->material_id; is.light_id = prim->light_id;
        }
    }
    /* the triangles continue the scan with the indices behind the spheres': same strict rule */
    for (uint32_t k = 0; k < sc->n_tris; ++k) {
        float thit;
        const ptor_triangle *tr = &sc->tris[k];
        if ((thit = triangle_intersect(tr, o, d, tmin, tmax)) > 0.f) {
            const v3 v0 = v3_load(tr->v0);
            tmax = thit;
            found = 1;
            is.prim = (int)(sc->n_prims + k);
            is.p = v3_add(o, v3_scale(d, thit));
            is.n = v3_normalize(v3_cross(v3_sub(v3_load(tr->v1), v0), v3_sub(v3_load(tr->v2), v0)));
            is.t = thit;
            is.material_id = tr->material_id; is.light_id = tr->light_id;
        }
    }
    *isect = is;
    return found;
}

/* ConcentricSampleDisk, montecarlo.h:76-118 (sinf/cosf replaced by the deterministic pair). */
static void concentric_sample_disk(float u1, float u2, float *dx, float *dy)
{
    float r, theta;
    float sx = 2 * u1 - 1;
    float sy = 2 * u2 - 1;
    if (sx == 0.f && sy == 0.f) { *dx = 0.f; *dy = 0.f; return; }
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
    ptor_sincosf(theta, &s, &c);
    *dx = r * c;
    *dy = r * s;
}

/* RotateByNormal, montecarlo.h:120-125: the tangent u is deliberately left un-normalised. */
static v3 rotate_by_normal(v3 wo, v3 n)
{
    v3 w = n;
    v3 u = fabsf(n.x) > fabsf(n.z) ? v3_make(-n.y, n.x, 0.f) : v3_make(0.f, -n.z, n.y);
    v3 v = v3_cross(w, u);
    return v3_normalize(v3_add(v3_add(v3_scale(u, wo.x), v3_scale(v, wo.y)), v3_scale(w, wo.z)));
}

/* CosineSampleHemisphere, montecarlo.h:127-133; max() is the ternary macro of globals.h:67. */
static v3 cosine_sample_hemisphere(float u1, float u2, v3 n)
{
    v3 ret = v3_make(0, 0, 0);
    concentric_sample_disk(u1, u2, &ret.x, &ret.y);
    float e = 1.f - ret.x * ret.x - ret.y * ret.y;
    ret.z = sqrtf(0.f > e ? 0.f : e);
    return rotate_by_normal(ret, n);
}

/* Reflect, globals.h:107-109 */
static v3 reflect(v3 wo, v3 n) { return v3_sub(wo, v3_scale(n, 2.f * v3_dot(wo, n))); }

/* Refract, globals.h:111-115 */
static v3 refract(v3 wo, v3 n, float eta)
{
    float cost = v3_dot(wo, n);
    float cos2t = 1.f - eta * eta * (1.f - cost * cost);
    return v3_normalize(v3_sub(v3_scale(wo, eta), v3_scale(n, eta * cost + sqrtf(cos2t))));
}

/* reflectance (Fresnel), globals.h:117-126 */
static float reflectance(v3 inc, v3 nor, float n1, float n2)
{
    float n = n1 / n2;
    float cosI = -(v3_dot(nor, inc));
    float sinT2 = n * n * (1.0f - cosI * cosI);
    if (sinT2 > 1.0f) return 1.0f;
    float cosT = sqrtf(1.0f - sinT2);
    float rOrth = (n1 * cosI - n2 * cosT) / (n1 * cosI + n2 * cosT);
    float rPar = (n2 * cosI - n1 * cosT) / (n2 * cosI + n1 * cosT);
    return (rOrth * rOrth + rPar * rPar) / 2.0f;
}

/* SampleMaterial, scene.h:177-221, with Material::F (material.h:37-43) inlined. */
static v3 sample_material(const ptor_material *mat, v3 wo, v3 *wi, float *pdf, v3 n, float su, float sv)
{
    v3 f = v3_make(0.f, 0.f, 0.f);
    v3 color = v3_load(mat->color);
    if (mat->type == 0) {
        *wi = cosine_sample_hemisphere(su, sv, n);
        *pdf = v3_dot(wo, *wi) < 0.f ? fabsf(v3_dot(*wi, n)) * PT_INV_PI : 0.f;
        f = v3_add(f, v3_scale(color, PT_INV_PI));
    } else if (mat->type == 1) {
        *wi = reflect(wo, n);
        *pdf = 1.f;
        f = v3_scale(color, mat->coef);
    } else if (mat->type == 2) {
        float n1, n2;
        v3 nnor;
        if (v3_dot(wo, n) < 0.f) { n1 = 1.f; n2 = mat->coef; nnor = n; }
        else                     { n1 = mat->coef; n2 = 1.f; nnor = v3_neg(n); }
        float refl = reflectance(wo, nnor, n1, n2);
        if (su < refl) *wi = reflect(wo, nnor);
        else           *wi = refract(wo, nnor, n1 / n2);
        f = color;
        *pdf = 1.f;
    } else {
        *wi = v3_make(0, 0, 0);
        *pdf = 0.f;
    }
    return f;
}

/* Trace, pathtracer.cu:112-170.  `hits` (may be NULL) receives max_bounces+1 entries. */
static v3 trace(const scene_ctx *sc, v3 o, v3 d, uint32_t pixel, uint32_t iteration,
                int32_t *hits, uint64_t *n_rays)
{
    v3 T = v3_make(1.f, 1.f, 1.f), L = v3_make(0.f, 0.f, 0.f);
    isect_t isect;
    int k = 0;
    const int mb = sc->max_bounces;
    if (hits) for (int i = 0; i <= mb; ++i) hits[i] = PTOR_HIT_NOT_TRACED;

    ++*n_rays;
    int found = scene_intersect(sc, o, d, &isect);
    if (hits) hits[k] = found ? isect.prim : PTOR_HIT_MISS;
    ++k;
    if (!found) return v3_make(0.f, 0.f, 0.f);

    isect_t isectp = isect;
    for (int bounces = 0;; ++bounces) {
        const ptor_material *mat = &sc->mats[isectp.material_id];
        const v3 p = isectp.p;
        const v3 n = isectp.n;
        const v3 wo = d;

        if (isectp.light_id > -1) {
            /* Light::L, light.h:43-45 */
            const ptor_light *light = &sc->lights[isectp.light_id];
            v3 le = v3_dot(n, v3_neg(wo)) > 0.f ? v3_load(light->intensity) : v3_make(0.f, 0.f, 0.f);
            L = v3_add(L, v3_mul(T, le));
        }

        float rnd[4];
        ptor_rng4(sc->seed, pixel, iteration, (uint32_t)bounces, PTOR_STREAM_PATH, rnd);

        v3 wi;
        float pdf;
        const v3 f = sample_material(mat, wo, &wi, &pdf, n, rnd[0], rnd[1]);
        if ((f.x == 0.f && f.y == 0.f && f.z == 0.f) || pdf == 0.f) break;

        /* pathThroughput *= f * fabs(wi.Dot(n)) / pdf   (pathtracer.cu:149) */
        T = v3_mul(T, v3_div(v3_scale(f, fabsf(v3_dot(wi, n))), pdf));
        o = p; d = wi;

        if (bounces > 3) {
            float gb = T.y > T.z ? T.y : T.z;      /* Color::Max, geometry.h:474 */
            float mx = T.x > gb ? T.x : gb;
            float q = .5f < mx ? .5f : mx;         /* min macro, globals.h:66 */
            if (rnd[2] > q) break;
            T = v3_div(T, q);
        }
        if (bounces == mb) break;

        ++*n_rays;
        found = scene_intersect(sc, o, d, &isect);
        if (hits) hits[k] = found ? isect.prim : PTOR_HIT_MISS;
        ++k;
        if (!found) break;
        isectp = isect;
    }
    return L;
}

#include "pt_oracle_nee.h"

void ptor_camera_init(ptor_camera *out, const float eye[3], const float look_at[3],
                      const float up_[3], uint32_t width, uint32_t height, float fov)
{
    /* camera.h:31-57 */
    const float aspect = (float)height / (float)width;
    const float hfov = fov;
    const float vfov = hfov * aspect;
    const v3 e = v3_load(eye), up = v3_load(up_);
    const v3 lookAt = v3_sub(v3_load(look_at), e);
    const float dist = v3_length(lookAt);
    const v3 w = v3_div(lookAt, dist);
    const v3 v = v3_normalize(v3_sub(up, v3_scale(w, v3_dot(up, w))));
    const v3 u = v3_normalize(v3_cross(w, v));
    /* camera.h:144-158; Radians is globals.h:95-97 */
    float magnitude = dist * 2.f * tanf((PT_M_PI / 180.f) * (hfov * 0.5f)) / (float)width;
    const v3 pxX = v3_scale(u, magnitude);
    magnitude = dist * 2.f * tanf((PT_M_PI / 180.f) * (vfov * 0.5f)) / (float)height;
    const v3 pxY = v3_scale(v, magnitude);
    const v3 offset = v3_sub(v3_scale(pxY, (float)((float)height * 0.5f)),
                             v3_scale(pxX, (float)((float)width * 0.5f)));
    const v3 first = v3_add(v3_scale(w, dist), offset);
    out->pos[0] = e.x; out->pos[1] = e.y; out->pos[2] = e.z;
    out->px_x[0] = pxX.x; out->px_x[1] = pxX.y; out->px_x[2] = pxX.z;
    out->px_y[0] = pxY.x; out->px_y[1] = pxY.y; out->px_y[2] = pxY.z;
    out->first_ray_dir[0] = first.x; out->first_ray_dir[1] = first.y; out->first_ray_dir[2] = first.z;
    out->width = width; out->height = height;
}

/* Camera::UpdateParameter, camera.h:144-158 */
static void camera_update_parameter(ptor_camera_state *st)
{
    const v3 u = v3_load(st->view[0]), v = v3_load(st->view[1]), w = v3_load(st->view[2]);
    const uint32_t width = st->cam.width, height = st->cam.height;
    float magnitude = st->dist * 2.f * tanf((PT_M_PI / 180.f) * (st->hfov * 0.5f)) / (float)width;
    const v3 pxX = v3_scale(u, magnitude);
    magnitude = st->dist * 2.f * tanf((PT_M_PI / 180.f) * (st->vfov * 0.5f)) / (float)height;
    const v3 pxY = v3_scale(v, magnitude);
    const v3 offset = v3_sub(v3_scale(pxY, (float)((float)height * 0.5f)),
                             v3_scale(pxX, (float)((float)width * 0.5f)));
    const v3 first = v3_add(v3_scale(w, st->dist), offset);
    st->cam.px_x[0] = pxX.x; st->cam.px_x[1] = pxX.y; st->cam.px_x[2] = pxX.z;
    st->cam.px_y[0] = pxY.x; st->cam.px_y[1] = pxY.y; st->cam.px_y[2] = pxY.z;
    st->cam.first_ray_dir[0] = first.x; st->cam.first_ray_dir[1] = first.y; st->cam.first_ray_dir[2] = first.z;
}

void ptor_camera_state_init(ptor_camera_state *st, const float eye[3], const float look_at[3],
                            const float up_[3], uint32_t width, uint32_t height, float fov)
{
    /* camera.h:31-57 */
    const float aspect = (float)height / (float)width;
    st->hfov = fov;
    st->vfov = st->hfov * aspect;
    const v3 e = v3_load(eye), up = v3_load(up_);
    const v3 lookAt = v3_sub(v3_load(look_at), e);
    st->dist = v3_length(lookAt);
    const v3 w = v3_div(lookAt, st->dist);
    const v3 v = v3_normalize(v3_sub(up, v3_scale(w, v3_dot(up, w))));
    const v3 u = v3_normalize(v3_cross(w, v));
    st->view[0][0] = u.x; st->view[0][1] = u.y; st->view[0][2] = u.z;
    st->view[1][0] = v.x; st->view[1][1] = v.y; st->view[1][2] = v.z;
    st->view[2][0] = w.x; st->view[2][1] = w.y; st->view[2][2] = w.z;
    st->cam.pos[0] = e.x; st->cam.pos[1] = e.y; st->cam.pos[2] = e.z;
    st->cam.width = width; st->cam.height = height;
    camera_update_parameter(st);
}

/* Camera::Translate, camera.h:79-90 */
void ptor_camera_translate(ptor_camera_state *st, const float delta[3])
{
    const v3 u = v3_load(st->view[0]), v = v3_load(st->view[1]), w = v3_load(st->view[2]);
    const v3 step = v3_add(v3_add(v3_scale(u, delta[0]), v3_scale(v, delta[1])), v3_scale(w, delta[2]));
    st->cam.pos[0] += step.x; st->cam.pos[1] += step.y; st->cam.pos[2] += step.z;
}

/* Matrix::operator*, geometry.h:249-258 */
static void mat4_mul(const float a[4][4], const float b[4][4], float r[4][4])
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j] + a[i][3] * b[3][j];
}

/* Camera::Rotate, camera.h:97-129 */
void ptor_camera_rotate(ptor_camera_state *st, const float theta[3])
{
    const float ctx = theta[0] == 0.f ? 1.f : cosf(theta[0]);
    const float cty = theta[1] == 0.f ? 1.f : cosf(theta[1]);
    const float stx = theta[0] == 0.f ? 0.f : sinf(theta[0]);
    const float sty = theta[1] == 0.f ? 0.f : sinf(theta[1]);
    const float rx[4][4] = { {1.f, 0.f, 0.f, 0.f}, {0.f, cty, -sty, 0.f}, {0.f, sty, cty, 0.f}, {0.f, 0.f, 0.f, 1.f} };
    const float ry[4][4] = { {ctx, 0.f, stx, 0.f}, {0.f, 1.f, 0.f, 0.f}, {-stx, 0.f, ctx, 0.f}, {0.f, 0.f, 0.f, 1.f} };
    float view[4][4], out[4][4], rxy[4][4];
    memset(view, 0, sizeof view);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) view[i][j] = st->view[i][j];
    view[3][3] = 1.f;
    if (theta[0] == 0.f) mat4_mul(rx, view, out);
    else if (theta[1] == 0.f) mat4_mul(ry, view, out);
    else { mat4_mul(rx, ry, rxy); mat4_mul(rxy, view, out); }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) st->view[i][j] = out[i][j];
    camera_update_parameter(st);
}

void ptor_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    ptor_philox4x32_10(ctr, key, out);
}

void ptor_uniforms(uint64_t seed, uint32_t pixel, uint32_t iteration, uint32_t slot,
                   uint32_t stream, float out[4])
{
    ptor_rng4(seed, pixel, iteration, slot, stream, out);
}

void ptor_sincos(float x, float *s, float *c) { ptor_sincosf(x, s, c); }

typedef struct {
    const scene_ctx *sc;
    uint32_t first_iteration, n_iterations;
    uint32_t x0, y0, x1, y1;
    uint32_t row_begin, row_end; /* this worker's rows */
    int accum_mode;
    float *buffer;
    int32_t *hit_ids;
    float *path_radiance;
    uint64_t n_rays;
} job_t;

/* One pixel of one iteration: GenerateRayPool + RenderKernel, pathtracer.cu:33-54,62-110. */
static void render_pixel(job_t *job, uint32_t x, uint32_t y, uint32_t it_idx)
{
    const scene_ctx *sc = job->sc;
    const ptor_camera *cam = sc->cam;
    const uint32_t iteration = job->first_iteration + it_idx;
    const uint32_t index = y * cam->width + x;
    const int mb = sc->max_bounces;
    const uint32_t rw = job->x1 - job->x0, rh = job->y1 - job->y0;
    const size_t rp = (size_t)it_idx * rw * rh + (size_t)(y - job->y0) * rw + (x - job->x0);

    /* CameraSamples, pathtracer.cu:34-48 */
    float s[8];
    ptor_rng4(sc->seed, index, iteration, 0, PTOR_STREAM_CAMERA, s);
    ptor_rng4(sc->seed, index, iteration, 1, PTOR_STREAM_CAMERA, s + 4);
    s[0] *= -0.5f; s[1] *= -0.5f;
    s[2] *= 0.5f;  s[3] *= -0.5f;
    s[4] *= -0.5f; s[5] *= 0.5f;
    s[6] *= 0.5f;  s[7] *= 0.5f;

    const v3 first = v3_load(cam->first_ray_dir), pxX = v3_load(cam->px_x), pxY = v3_load(cam->px_y);
    v3 c = v3_make(0.f, 0.f, 0.f);
    for (int i = 0; i < 4; ++i) {
        /* Camera::GenerateRay, camera.h:66-72 */
        float sx = ((float)(int)x + s[i * 2]);
        float sy = ((float)(int)y + s[i * 2 + 1]);
        v3 dir = v3_normalize(v3_add(v3_sub(first, v3_scale(pxY, sy)), v3_scale(pxX, sx)));
        int32_t *hits = job->hit_ids ? job->hit_ids + (rp * 4 + i) * (size_t)(mb + 1) : NULL;
        v3 Lp = sc->integrator == PTOR_INTEGRATOR_NEE_MIS
                    ? trace_nee(sc, v3_load(cam->pos), dir, index, iteration, hits, &job->n_rays)
                    : trace(sc, v3_load(cam->pos), dir, index, iteration, hits, &job->n_rays);
        if (job->path_radiance) {
            float *pr = job->path_radiance + (rp * 4 + i) * 3;
            pr[0] = Lp.x; pr[1] = Lp.y; pr[2] = Lp.z;
        }
        c = v3_add(c, Lp);
    }
    c = v3_div(c, 4.f);

    float *px = job->buffer + (size_t)index * 3;
    if (job->accum_mode == PTOR_ACCUM_RUNNING_MEAN) {
        float invIteration = 1.f / (float)(iteration);
        px[0] = (px[0] * (float)(iteration - 1) + c.x) * invIteration;
        px[1] = (px[1] * (float)(iteration - 1) + c.y) * invIteration;
        px[2] = (px[2] * (float)(iteration - 1) + c.z) * invIteration;
    } else {
        px[0] += c.x; px[1] += c.y; px[2] += c.z;
    }
}

static void *worker(void *arg)
{
    job_t *job = (job_t *)arg;
    for (uint32_t it = 0; it < job->n_iterations; ++it)
        for (uint32_t y = job->row_begin; y < job->row_end; ++y)
            for (uint32_t x = job->x0; x < job->x1; ++x)
                render_pixel(job, x, y, it);
    return NULL;
}

/* Test hook: SampleMaterial (scene.h:177-221) row by row.  in: [n][8] = wo.xyz, n.xyz, su, sv;
 * out: [n][8] = f.rgb, wi.xyz, pdf, 0. */
int ptor_sample_material_eval(const ptor_material *mat, const float *in, float *out, uint32_t n)
{
    if (!mat || !in || !out) return -1;
    for (uint32_t i = 0; i < n; ++i) {
        const float *x = in + 8 * (size_t)i;
        float *y = out + 8 * (size_t)i;
        v3 wi = v3_make(0.f, 0.f, 0.f);
        float pdf = 0.f;
        const v3 f = sample_material(mat, v3_make(x[0], x[1], x[2]), &wi, &pdf, v3_make(x[3], x[4], x[5]), x[6], x[7]);
        y[0] = f.x; y[1] = f.y; y[2] = f.z; y[3] = wi.x; y[4] = wi.y; y[5] = wi.z; y[6] = pdf; y[7] = 0.f;
    }
    return 0;
}

int ptor_render(const ptor_camera *cam,
                const ptor_material *mats, uint32_t n_mats,
                const ptor_sphere *prims, uint32_t n_prims,
                const ptor_light *lights, uint32_t n_lights,
                uint64_t seed, int max_bounces,
                uint32_t first_iteration, uint32_t n_iterations,
                uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                int accum_mode, float *buffer,
                int32_t *hit_ids, float *path_radiance, uint64_t *ray_count,
                int n_threads)
{
    return ptor_render_with(PTOR_INTEGRATOR_PATH, cam, mats, n_mats, prims, n_prims, lights, n_lights, seed,
                            max_bounces, first_iteration, n_iterations, x0, y0, x1, y1, accum_mode, buffer,
                            hit_ids, path_radiance, ray_count, n_threads);
}

int ptor_render_with(int integrator, const ptor_camera *cam,
                     const ptor_material *mats, uint32_t n_mats,
                     const ptor_sphere *prims, uint32_t n_prims,
                     const ptor_light *lights, uint32_t n_lights,
                     uint64_t seed, int max_bounces,
                     uint32_t first_iteration, uint32_t n_iterations,
                     uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                     int accum_mode, float *buffer,
                     int32_t *hit_ids, float *path_radiance, uint64_t *ray_count,
                     int n_threads)
{
    return ptor_render_tri(integrator, cam, mats, n_mats, prims, n_prims, NULL, 0, lights, n_lights, seed, max_bounces,
                           first_iteration, n_iterations, x0, y0, x1, y1, accum_mode, buffer, hit_ids, path_radiance,
                           ray_count, n_threads);
}

int ptor_render_tri(int integrator, const ptor_camera *cam,
                    const ptor_material *mats, uint32_t n_mats,
                    const ptor_sphere *prims, uint32_t n_prims,
                    const ptor_triangle *tris, uint32_t n_tris,
                    const ptor_light *lights, uint32_t n_lights,
                    uint64_t seed, int max_bounces,
                    uint32_t first_iteration, uint32_t n_iterations,
                    uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                    int accum_mode, float *buffer,
                    int32_t *hit_ids, float *path_radiance, uint64_t *ray_count,
                    int n_threads)
{
    if (integrator != PTOR_INTEGRATOR_PATH && integrator != PTOR_INTEGRATOR_NEE_MIS) return -1;
    if (n_tris && (!tris || integrator != PTOR_INTEGRATOR_PATH)) return -1;
    for (uint32_t i = 0; i < n_tris; ++i)
        if (tris[i].material_id >= n_mats || tris[i].light_id >= (int32_t)n_lights) return -1;
    for (uint32_t i = 0; i < n_lights; ++i)
        if (lights && lights[i].type == 1 && lights[i].u.prim_id >= n_prims + n_tris) return -1;
    if (!cam || !buffer || max_bounces < 0 || first_iteration == 0) return -1;
    if (x1 > cam->width || y1 > cam->height || x0 > x1 || y0 > y1) return -1;
    if ((n_prims && !prims) || (n_mats && !mats) || (n_lights && !lights)) return -1;
    for (uint32_t i = 0; i < n_prims; ++i) {
        if (prims[i].material_id >= n_mats) return -1;
        if (prims[i].light_id >= (int32_t)n_lights) return -1;
    }
    scene_ctx sc = { cam, mats, prims, lights, n_mats, n_prims, n_lights, seed, max_bounces, integrator, tris, n_tris };

    uint32_t rows = y1 - y0;
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > rows) n_threads = rows ? (int)rows : 1;
    job_t *jobs = (job_t *)calloc((size_t)n_threads, sizeof(job_t));
    pthread_t *tids = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    if (!jobs || !tids) { free(jobs); free(tids); return -1; }
    for (int t = 0; t < n_threads; ++t) {
        job_t *j = &jobs[t];
        j->sc = &sc;
        j->first_iteration = first_iteration; j->n_iterations = n_iterations;
        j->x0 = x0; j->y0 = y0; j->x1 = x1; j->y1 = y1;
        j->row_begin = y0 + (uint32_t)(((uint64_t)rows * (uint32_t)t) / (uint32_t)n_threads);
        j->row_end = y0 + (uint32_t)(((uint64_t)rows * (uint32_t)(t + 1)) / (uint32_t)n_threads);
        j->accum_mode = accum_mode;
        j->buffer = buffer; j->hit_ids = hit_ids; j->path_radiance = path_radiance;
        j->n_rays = 0;
    }
    if (n_threads == 1) {
        worker(&jobs[0]);
    } else {
        for (int t = 0; t < n_threads; ++t) pthread_create(&tids[t], NULL, worker, &jobs[t]);
        for (int t = 0; t < n_threads; ++t) pthread_join(tids[t], NULL);
    }
    uint64_t total = 0;
    for (int t = 0; t < n_threads; ++t) total += jobs[t].n_rays;
    if (ray_count) *ray_count = total;
    free(jobs); free(tids);
    return 0;
}
