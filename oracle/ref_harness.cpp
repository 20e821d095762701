/*
 * TEST INFRASTRUCTURE ONLY -- host build of the REFERENCE's own trace logic.
 *
 * This translation unit contains no restated algorithm: it #includes the reference headers
 * where they lie (-I$(REFERENCE), normally /root/reference) and the function bodies that
 * oracle/Makefile cuts out of the reference at build time into oracle/_ref/ (never
 * committed):
 *     scene_host.inc   = scene.h with its single `#include <cuda_runtime.h>` line dropped
 *                        (nothing from that header is used when the code is compiled for host)
 *     camsamples.inc   = pathtracer.cu:33-54   (struct CameraSamples)
 *     trace.inc        = pathtracer.cu:112-170 (Trace)
 * The rest of pathtracer.cu needs nvcc, cuRAND and OpenGL and cannot be built here, so the
 * two kernels' bodies (pathtracer.cu:62-75,80-110) are driven by the loop in ref_render().
 *
 * It can only be BUILT where the reference tree exists (the development container).  The built
 * libraries (oracle/_ref/*.so: git-ignored, so never in the history, but not excluded from what
 * is sent to the GPU box) travel with the tree like every other built library: on the GPU box
 * bench.py times libpt_ref.so on the host cores as the cpu_baseline's `reference_host_build`
 * leg -- the reference's CPU path beside the GPU number -- and nothing else there uses it.
 * The tests there use the golden vectors this harness produced (oracle/gen_golden.py); no
 * reference source and no text cut from it leaves the container.
 *
 * What is NOT the reference here, and why:
 *   - the random floats: the reference reads a cuRAND buffer seeded with time(NULL); the
 *     harness fills the same buffer layout from the Philox streams of pt_oracle_math.h;
 *   - with PT_REF_INJECT_SINCOS, sinf/cosf inside the reference headers resolve to the
 *     deterministic pair of pt_oracle_math.h (libm and the GPU differ in the last bit); the
 *     build without that macro keeps libm and is used to bound the effect on the image;
 *   - all pixels are rendered, not only floor(W/16)*16 x floor(H/16)*16 (pathtracer.cu:229).
 *
 * Must be compiled with clang++ (g++ rejects light.h:49), -DNDEBUG (geometry.h:330 asserts
 * on an undeclared name) and -ffp-contract=off.
 */
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <cfloat>
#include <cstdio>
#include <math.h>
#include <new>
#include <vector>

#include "pt_oracle.h"
#include "pt_oracle_math.h"

#ifdef PT_REF_INJECT_SINCOS
#define sinf pt_sinf
#define cosf pt_cosf
#endif

#define __device__
#define __global__
/* Camera keeps pxX / pxY / firstRayDir and the view state private and has no accessors for them
 * (camera.h:159-171); the camera fixtures compare exactly those members, so the one header is
 * included with `private` spelled `public`.  Access control does not change layout or code. */
#define private public
#include "camera.h"
#undef private
#include "material.h"
#include "primitive.h"
#include "light.h"
#include "_ref/scene_host.inc"

static thread_local int32_t *g_hits = nullptr;
static thread_local int g_hit_k = 0;
static thread_local uint64_t g_rays = 0;

static inline bool HookedIntersect(const Scene *scene, const Ray &ray, Intersection *isect)
{
    bool found = Intersect(scene, ray, isect);
    ++g_rays;
    if (g_hits)
        g_hits[g_hit_k] = found ? (int32_t)(isect->prim - scene->primitives.primitives) : PTOR_HIT_MISS;
    ++g_hit_k;
    return found;
}

#define Intersect(...) HookedIntersect(__VA_ARGS__)
#include "_ref/camsamples.inc"
#include "_ref/trace.inc"
#undef Intersect

extern "C" {

/* sizeof table of the reference's device-visible structs. */
int ref_sizeof(int which)
{
    switch (which) {
    case 0: return (int)sizeof(Vec);
    case 1: return (int)sizeof(Point);
    case 2: return (int)sizeof(Ray);
    case 3: return (int)sizeof(Color);
    case 4: return (int)sizeof(BBox);
    case 5: return (int)sizeof(Primitive);
    case 6: return (int)sizeof(Material);
    case 7: return (int)sizeof(Light);
    case 8: return (int)sizeof(Camera);
    case 9: return (int)sizeof(Scene);
    case 10: return (int)sizeof(Intersection);
    default: return -1;
    }
}

int ref_uses_libm_sincos(void)
{
#ifdef PT_REF_INJECT_SINCOS
    return 0;
#else
    return 1;
#endif
}

void ref_camera_init(ptor_camera *out, const float eye[3], const float look_at[3],
                     const float up[3], uint32_t width, uint32_t height, float fov)
{
    Camera cam(Point(eye[0], eye[1], eye[2]), Point(look_at[0], look_at[1], look_at[2]),
               Vec(up[0], up[1], up[2]), width, height, 0.f, 0.f, fov);
    out->pos[0] = cam.pos.x; out->pos[1] = cam.pos.y; out->pos[2] = cam.pos.z;
    out->px_x[0] = cam.pxX.x; out->px_x[1] = cam.pxX.y; out->px_x[2] = cam.pxX.z;
    out->px_y[0] = cam.pxY.x; out->px_y[1] = cam.pxY.y; out->px_y[2] = cam.pxY.z;
    out->first_ray_dir[0] = cam.firstRayDir.x; out->first_ray_dir[1] = cam.firstRayDir.y;
    out->first_ray_dir[2] = cam.firstRayDir.z;
    out->width = cam.Width(); out->height = cam.Height();
}

/* Applies n_ops moves to the reference's Camera: ops[i] = {kind (0 Translate, 1 Rotate), x, y, z};
 * pods[i] receives the derived vectors after move i (pods[0] is the state before any move, so
 * pods has n_ops + 1 entries). */
void ref_camera_moves(const float eye[3], const float look_at[3], const float up[3], uint32_t width,
                      uint32_t height, float fov, const float *ops, uint32_t n_ops, ptor_camera *pods)
{
    Camera cam(Point(eye[0], eye[1], eye[2]), Point(look_at[0], look_at[1], look_at[2]),
               Vec(up[0], up[1], up[2]), width, height, 0.f, 0.f, fov);
    for (uint32_t i = 0; i <= n_ops; ++i) {
        if (i > 0) {
            const float *op = ops + (size_t)(i - 1) * 4;
            if (op[0] == 0.f) cam.Translate(Vec(op[1], op[2], op[3]));
            else cam.Rotate(Vec(op[1], op[2], op[3]));
        }
        ptor_camera *out = &pods[i];
        out->pos[0] = cam.pos.x; out->pos[1] = cam.pos.y; out->pos[2] = cam.pos.z;
        out->px_x[0] = cam.pxX.x; out->px_x[1] = cam.pxX.y; out->px_x[2] = cam.pxX.z;
        out->px_y[0] = cam.pxY.x; out->px_y[1] = cam.pxY.y; out->px_y[2] = cam.pxY.z;
        out->first_ray_dir[0] = cam.firstRayDir.x; out->first_ray_dir[1] = cam.firstRayDir.y;
        out->first_ray_dir[2] = cam.firstRayDir.z;
        out->width = cam.Width(); out->height = cam.Height();
    }
}

/* Same contract as ptor_render(), but the camera is built by the reference's constructor. */
int ref_render(const float eye[3], const float look_at[3], const float up[3],
               uint32_t width, uint32_t height, float fov,
               const ptor_material *mats, uint32_t n_mats,
               const ptor_sphere *prims, uint32_t n_prims,
               const ptor_light *lights, uint32_t n_lights,
               uint64_t seed, int max_bounces,
               uint32_t first_iteration, uint32_t n_iterations,
               uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
               int accum_mode, float *buffer,
               int32_t *hit_ids, float *path_radiance, uint64_t *ray_count)
{
    if (!buffer || max_bounces < 0 || first_iteration == 0) return -1;
    if (x1 > width || y1 > height || x0 > x1 || y0 > y1) return -1;

    Camera camera(Point(eye[0], eye[1], eye[2]), Point(look_at[0], look_at[1], look_at[2]),
                  Vec(up[0], up[1], up[2]), width, height, 0.f, 0.f, fov);

    std::vector<Material> mv;
    for (uint32_t i = 0; i < n_mats; ++i) {
        Color c(mats[i].color[0], mats[i].color[1], mats[i].color[2]);
        if (mats[i].type == 0) mv.push_back(CreateDiffuseMaterial(c, mats[i].coef));
        else if (mats[i].type == 1) mv.push_back(CreateSpecularMaterial(c, mats[i].coef));
        else mv.push_back(CreateTransmissiveMaterial(c, mats[i].coef));
    }
    std::vector<Primitive> pv;
    for (uint32_t i = 0; i < n_prims; ++i)
        pv.push_back(Primitive(Point(prims[i].pos[0], prims[i].pos[1], prims[i].pos[2]),
                               prims[i].radius, prims[i].material_id, prims[i].light_id));
    /* Light has a deleted copy constructor under standard C++ (light.h:48-51): build in place. */
    Light *lv = (Light *)malloc(sizeof(Light) * (n_lights ? n_lights : 1));
    for (uint32_t i = 0; i < n_lights; ++i) {
        Color c(lights[i].intensity[0], lights[i].intensity[1], lights[i].intensity[2]);
        if (lights[i].type == 1) new (&lv[i]) Light(lights[i].u.prim_id, c);
        else new (&lv[i]) Light(Point(lights[i].u.pos[0], lights[i].u.pos[1], lights[i].u.pos[2]), c);
    }
    Scene scene(MaterialList(mv.data(), (uint32_t)mv.size()),
                PrimitiveList(pv.data(), (uint32_t)pv.size()), LightList(lv, n_lights));

    const int mb = max_bounces;
    const uint32_t rw = x1 - x0, rh = y1 - y0;
    std::vector<float> rng((size_t)(mb + 1) * 3);
    g_rays = 0;

    for (uint32_t it_idx = 0; it_idx < n_iterations; ++it_idx) {
        const uint32_t iteration = first_iteration + it_idx;
        for (uint32_t y = y0; y < y1; ++y)
            for (uint32_t x = x0; x < x1; ++x) {
                const uint32_t index = y * width + x;
                const size_t rp = (size_t)it_idx * rw * rh + (size_t)(y - y0) * rw + (x - x0);

                /* GenerateRayPool body, pathtracer.cu:70-74 */
                float cs[8];
                ptor_rng4(seed, index, iteration, 0, PTOR_STREAM_CAMERA, cs);
                ptor_rng4(seed, index, iteration, 1, PTOR_STREAM_CAMERA, cs + 4);
                CameraSamples samples(cs);
                Ray rays[4];
                for (int i = 0; i < 4; ++i) rays[i] = camera.GenerateRay(x, y, samples[i]);

                /* RenderKernel body, pathtracer.cu:92-109 */
                for (int b = 0; b <= mb; ++b) {
                    float r4[4];
                    ptor_rng4(seed, index, iteration, (uint32_t)b, PTOR_STREAM_PATH, r4);
                    rng[(size_t)b * 3 + 0] = r4[0];
                    rng[(size_t)b * 3 + 1] = r4[1];
                    rng[(size_t)b * 3 + 2] = r4[2];
                }
                Color c;
                for (int i = 0; i < 4; ++i) {
                    Ray ray = rays[i];
                    int32_t *hits = hit_ids ? hit_ids + (rp * 4 + i) * (size_t)(mb + 1) : nullptr;
                    if (hits) for (int k = 0; k <= mb; ++k) hits[k] = PTOR_HIT_NOT_TRACED;
                    g_hits = hits; g_hit_k = 0;
                    Color Lp = Trace(ray, &scene, rng.data(), mb);
                    g_hits = nullptr;
                    if (path_radiance) {
                        float *pr = path_radiance + (rp * 4 + i) * 3;
                        pr[0] = Lp.r; pr[1] = Lp.g; pr[2] = Lp.b;
                    }
                    c += Lp;
                }
                c /= 4.f;

                const uint32_t fbindex = index * 3;
                if (accum_mode == PTOR_ACCUM_RUNNING_MEAN) {
                    float invIteration = 1.f / (float)(iteration);
                    buffer[fbindex]     = (buffer[fbindex]     * (float)(iteration - 1) + c.r) * invIteration;
                    buffer[fbindex + 1] = (buffer[fbindex + 1] * (float)(iteration - 1) + c.g) * invIteration;
                    buffer[fbindex + 2] = (buffer[fbindex + 2] * (float)(iteration - 1) + c.b) * invIteration;
                } else {
                    buffer[fbindex] += c.r; buffer[fbindex + 1] += c.g; buffer[fbindex + 2] += c.b;
                }
            }
    }
    if (ray_count) *ray_count = g_rays;
    free(lv);
    return 0;
}

/* The reference's direct-light sampling functions, which Trace never calls, evaluated one by one
 * exactly as written (same contract as ptor_nee_eval with as_written = 1). */
int ref_nee_eval(int op, const ptor_material *mats, uint32_t n_mats, const ptor_sphere *prims, uint32_t n_prims,
                 const ptor_light *lights, uint32_t n_lights, const float *in_all, float *out_all, uint32_t count)
{
    std::vector<Material> mv;
    for (uint32_t i = 0; i < n_mats; ++i) {
        Color c(mats[i].color[0], mats[i].color[1], mats[i].color[2]);
        if (mats[i].type == 0) mv.push_back(CreateDiffuseMaterial(c, mats[i].coef));
        else if (mats[i].type == 1) mv.push_back(CreateSpecularMaterial(c, mats[i].coef));
        else mv.push_back(CreateTransmissiveMaterial(c, mats[i].coef));
    }
    std::vector<Primitive> pv;
    for (uint32_t i = 0; i < n_prims; ++i)
        pv.push_back(Primitive(Point(prims[i].pos[0], prims[i].pos[1], prims[i].pos[2]),
                               prims[i].radius, prims[i].material_id, prims[i].light_id));
    Light *lv = (Light *)malloc(sizeof(Light) * (n_lights ? n_lights : 1));
    for (uint32_t i = 0; i < n_lights; ++i) {
        Color c(lights[i].intensity[0], lights[i].intensity[1], lights[i].intensity[2]);
        if (lights[i].type == 1) new (&lv[i]) Light(lights[i].u.prim_id, c);
        else new (&lv[i]) Light(Point(lights[i].u.pos[0], lights[i].u.pos[1], lights[i].u.pos[2]), c);
    }
    Scene scene(MaterialList(mv.data(), (uint32_t)mv.size()),
                PrimitiveList(pv.data(), (uint32_t)pv.size()), LightList(lv, n_lights));
    int rc = 0;
    for (uint32_t i = 0; i < count && rc == 0; ++i) {
        const float *in = in_all + (size_t)i * 16;
        float *out = out_all + (size_t)i * 16;
        for (int k = 0; k < 16; ++k) out[k] = 0.f;
        auto put3 = [out](int at, float x, float y, float z) { out[at] = x; out[at + 1] = y; out[at + 2] = z; };
        switch (op) {
        case PTOR_NEE_SAMPLE_SPHERE: { Vec v = UniformSampleSphere(in[0], in[1]); put3(0, v.x, v.y, v.z); break; }
        case PTOR_NEE_SAMPLE_CONE: {
            Vec v = UniformSampleCone(in[0], in[1], in[2], Vec(in[3], in[4], in[5]), Vec(in[6], in[7], in[8]),
                                      Vec(in[9], in[10], in[11]));
            put3(0, v.x, v.y, v.z); break;
        }
        case PTOR_NEE_CONE_PDF: out[0] = UniformConePdf(in[0]); break;
        case PTOR_NEE_POWER_HEURISTIC: out[0] = PowerHeuristic((int)in[0], in[1], (int)in[2], in[3]); break;
        case PTOR_NEE_COORD_SYSTEM: {
            Vec v2, v3;
            CoordinateSystem(Vec(in[0], in[1], in[2]), &v2, &v3);
            put3(0, v2.x, v2.y, v2.z); put3(3, v3.x, v3.y, v3.z); break;
        }
        case PTOR_NEE_MATERIAL: {
            Material m;
            m.type = (MaterialType)(int)in[0]; m.color = Color(in[1], in[2], in[3]); m.coef = 0.f;
            const Vec wo(in[4], in[5], in[6]), wi(in[7], in[8], in[9]), n(in[10], in[11], in[12]);
            Color f = m.F(wo, wi, n);
            put3(0, f.r, f.g, f.b);
            out[3] = Pdf(&m, wo, wi, n); break;
        }
        case PTOR_NEE_LIGHT_L: {
            Light l(0u, Color(in[0], in[1], in[2]));
            Color c = l.L(Point(), Vec(in[3], in[4], in[5]), Vec(in[6], in[7], in[8]));
            put3(0, c.r, c.g, c.b); break;
        }
        case PTOR_NEE_SEGMENT: {
            VisibilityTester vt;
            vt.SetSegment(Point(in[0], in[1], in[2]), in[3], Point(in[4], in[5], in[6]), in[7]);
            put3(0, vt.r.o.x, vt.r.o.y, vt.r.o.z); put3(3, vt.r.d.x, vt.r.d.y, vt.r.d.z);
            out[6] = vt.mint; out[7] = vt.maxt; break;
        }
        case PTOR_NEE_SPHERE_SAMPLE_AREA: {
            Primitive prim(Point(in[0], in[1], in[2]), in[3], 0);
            Vec ns;
            Point p = prim.Sample(UVSample(in[4], in[5]), &ns);
            put3(0, p.x, p.y, p.z); put3(3, ns.x, ns.y, ns.z); break;
        }
        case PTOR_NEE_SPHERE_SAMPLE_FROM: {
            Primitive prim(Point(in[0], in[1], in[2]), in[3], 0);
            Vec ns;
            Point p = prim.Sample(Point(in[4], in[5], in[6]), UVSample(in[7], in[8]), &ns);
            put3(0, p.x, p.y, p.z); put3(3, ns.x, ns.y, ns.z); break;
        }
        case PTOR_NEE_LIGHT_PDF:
            if ((uint32_t)in[0] >= n_prims) { rc = -1; break; }
            out[0] = Pdf(&scene, scene.primitives[(uint32_t)in[0]], Point(in[1], in[2], in[3]), Vec(in[4], in[5], in[6]));
            break;
        case PTOR_NEE_INTERSECT_P:
            out[0] = IntersectP(&scene, Ray(Point(in[0], in[1], in[2]), Vec(in[3], in[4], in[5])), in[6], in[7]) ? 1.f : 0.f;
            break;
        case PTOR_NEE_SAMPLE_LIGHT: {
            if ((uint32_t)in[0] >= n_lights) { rc = -1; break; }
            Vec wi;
            float pdf = 0.f;
            VisibilityTester vt;
            Color li = SampleLight(scene.lights[(uint32_t)in[0]], Point(in[1], in[2], in[3]), &wi, &pdf, &vt,
                                   UVSample(in[4], in[5]), &scene);
            put3(0, li.r, li.g, li.b); put3(3, wi.x, wi.y, wi.z); out[6] = pdf;
            put3(7, vt.r.o.x, vt.r.o.y, vt.r.o.z); put3(10, vt.r.d.x, vt.r.d.y, vt.r.d.z);
            out[13] = vt.mint; out[14] = vt.maxt; break;
        }
        default: rc = -1; break;
        }
    }
    free(lv);
    return rc;
}

} /* extern "C" */
