/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the path-tracing hot path.  It is the checker,
 * never the thing shipped or measured: the product library (libptmi.so) does not include,
 * link or call anything in this directory.
 *
 * Plain-C restatement of the reference's per-pixel pipeline
 *   GenerateRayPool (pathtracer.cu:62-75) -> RenderKernel (pathtracer.cu:80-110)
 *   -> Trace (pathtracer.cu:112-170) -> Intersect / SampleMaterial (scene.h:71-94,177-221)
 * with every fp32 expression evaluated in the reference's order (no fused multiply-add).
 * Pinned against the reference's own headers compiled for host (oracle/_ref, built by
 * oracle/Makefile) through the golden vectors in tests/golden/.
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The four derived camera vectors GenerateRay reads (camera.h:66-72,165-171). */
typedef struct {
    float pos[3];
    float px_x[3];
    float px_y[3];
    float first_ray_dir[3];
    uint32_t width, height;
} ptor_camera;

/* material.h:45-47 */
typedef struct {
    float color[3];
    float coef;
    int32_t type; /* 0 DIFFUSE, 1 SPECULAR, 2 TRANSMISSIVE (material.h:25) */
} ptor_material;

/* primitive.h:78-81 without the never-read bounding box */
typedef struct {
    float pos[3];
    float radius;
    uint32_t material_id;
    int32_t light_id; /* -1: not a light */
} ptor_sphere;

/* A triangle primitive.  The reference has none -- primitive.h:25-28 says "Represents a sphere - will
 * be changed to triangle later" -- so nothing here can be pinned to it (PARITY UNPINNED): this is the
 * oracle of this build's own extension, stated in the reference's style.  Primitive k of the triangle
 * array has index n_prims + k in hit_ids and in Light::prim_id; the geometric normal
 * normalize((v1 - v0) x (v2 - v0)) plays the part of Primitive::Normal and is never flipped, like a
 * sphere's outward normal (pathtracer.cu:130, light.h:43-45). */
typedef struct {
    float v0[3], v1[3], v2[3];
    uint32_t material_id;
    int32_t light_id; /* -1: not a light */
} ptor_triangle;

/* light.h:47-52 */
typedef struct {
    int32_t type; /* 0 POINT_LIGHT, 1 AREA_LIGHT (light.h:34) */
    union {
        float pos[3];
        uint32_t prim_id;
    } u;
    float intensity[3];
} ptor_light;

enum { PTOR_ACCUM_RUNNING_MEAN = 0, PTOR_ACCUM_SUM = 1 };

/* hit_ids sentinel values */
enum { PTOR_HIT_MISS = -1, PTOR_HIT_NOT_TRACED = -2 };

/* Camera ctor + UpdateParameter, camera.h:31-57,144-158. */
void ptor_camera_init(ptor_camera *out, const float eye[3], const float look_at[3],
                      const float up[3], uint32_t width, uint32_t height, float fov_deg);

/* Interactive camera state and its two mutators (camera.h:79-129); `cam` is kept up to date. */
typedef struct {
    float view[3][3];
    float dist, hfov, vfov;
    ptor_camera cam;
} ptor_camera_state;
void ptor_camera_state_init(ptor_camera_state *st, const float eye[3], const float look_at[3],
                            const float up[3], uint32_t width, uint32_t height, float fov_deg);
void ptor_camera_translate(ptor_camera_state *st, const float delta[3]);
void ptor_camera_rotate(ptor_camera_state *st, const float theta[3]);

/* Philox4x32-10 block and the float map, exported for the known-answer tests. */
void ptor_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void ptor_uniforms(uint64_t seed, uint32_t pixel, uint32_t iteration, uint32_t slot,
                   uint32_t stream, float out[4]);
void ptor_sincos(float x, float *s, float *c);

/*
 * Render iterations first_iteration .. first_iteration+n_iterations-1 (1-based, the value
 * RenderKernel receives as `iteration`, pathtracer.cu:239) of the pixel rectangle
 * [x0,x1) x [y0,y1) into `buffer` (width*height*3 interleaved RGB, row 0 on top).
 *
 * accum_mode RUNNING_MEAN: buffer = (buffer*(it-1) + c) * (1/it)   (pathtracer.cu:104-109)
 *            SUM:          buffer += c
 *
 * Optional outputs (may be NULL), indexed [iteration][pixel of the rectangle, row-major]
 * [sub-sample 0..3]:
 *   hit_ids       [...][max_bounces+1] int32: primitive index returned by the k-th Intersect
 *                 call of the path, PTOR_HIT_MISS, or PTOR_HIT_NOT_TRACED
 *   path_radiance [...][3] float: the Color Trace returned
 * ray_count (may be NULL) receives the number of Intersect calls.
 * n_threads > 1 splits the rows over that many pthreads (results are identical).
 * Returns 0, or -1 on invalid arguments.
 */
int ptor_render(const ptor_camera *cam,
                const ptor_material *mats, uint32_t n_mats,
                const ptor_sphere *prims, uint32_t n_prims,
                const ptor_light *lights, uint32_t n_lights,
                uint64_t seed, int max_bounces,
                uint32_t first_iteration, uint32_t n_iterations,
                uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                int accum_mode, float *buffer,
                int32_t *hit_ids, float *path_radiance, uint64_t *ray_count,
                int n_threads);

/* Same with a choice of integrator: PATH is Trace as the reference runs it; NEE_MIS adds the
 * reference's dormant direct-light sampling (see pt_oracle_nee.h).  ray_count then includes the
 * shadow rays; hit_ids still lists the nearest-hit calls only. */
enum { PTOR_INTEGRATOR_PATH = 0, PTOR_INTEGRATOR_NEE_MIS = 1 };
int ptor_render_with(int integrator, const ptor_camera *cam,
                     const ptor_material *mats, uint32_t n_mats,
                     const ptor_sphere *prims, uint32_t n_prims,
                     const ptor_light *lights, uint32_t n_lights,
                     uint64_t seed, int max_bounces,
                     uint32_t first_iteration, uint32_t n_iterations,
                     uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                     int accum_mode, float *buffer,
                     int32_t *hit_ids, float *path_radiance, uint64_t *ray_count,
                     int n_threads);

/* The direct-light sampling functions one by one, for the pin tests: `count` items of 16 floats
 * in, 16 floats out (unused slots 0).  as_written selects the reference's text, bugs included,
 * where a repaired form exists (see pt_oracle_nee.h).  Ops 10-12 need the scene.
 *   op                         in                                   out
 *   SAMPLE_SPHERE              u1 u2                                dir(3)
 *   SAMPLE_CONE                u1 u2 cosThetaMax x(3) y(3) z(3)     dir(3)
 *   CONE_PDF                   cosThetaMax                          pdf
 *   POWER_HEURISTIC            nf fPdf ng gPdf                      weight
 *   COORD_SYSTEM               v1(3)                                v2(3) v3(3)
 *   MATERIAL                   type color(3) wo(3) wi(3) n(3)       F(3) Pdf
 *   LIGHT_L                    intensity(3) w(3) n(3)               L(3)
 *   SEGMENT                    p1(3) eps1 p2(3) eps2                o(3) d(3) mint maxt
 *   SPHERE_SAMPLE_AREA         pos(3) radius u v                    p(3) ns(3)
 *   SPHERE_SAMPLE_FROM         pos(3) radius p(3) u v               ps(3) ns(3)
 *   LIGHT_PDF                  prim p(3) wi(3)                      pdf
 *   INTERSECT_P                o(3) d(3) tmin tmax                  0 / 1
 *   SAMPLE_LIGHT               light p(3) u v                       Li(3) wi(3) pdf o(3) d(3) mint maxt */
enum {
    PTOR_NEE_SAMPLE_SPHERE = 0, PTOR_NEE_SAMPLE_CONE, PTOR_NEE_CONE_PDF, PTOR_NEE_POWER_HEURISTIC,
    PTOR_NEE_COORD_SYSTEM, PTOR_NEE_MATERIAL, PTOR_NEE_LIGHT_L, PTOR_NEE_SEGMENT,
    PTOR_NEE_SPHERE_SAMPLE_AREA, PTOR_NEE_SPHERE_SAMPLE_FROM, PTOR_NEE_LIGHT_PDF, PTOR_NEE_INTERSECT_P,
    PTOR_NEE_SAMPLE_LIGHT
};
/* SampleMaterial (scene.h:177-221) for n rows: in [n][8] = wo.xyz, n.xyz, su, sv; out [n][8] =
 * f.rgb, wi.xyz, pdf, 0. */
int ptor_sample_material_eval(const ptor_material *mat, const float *in, float *out, uint32_t n);

int ptor_nee_eval(int op, int as_written,
                  const ptor_material *mats, uint32_t n_mats, const ptor_sphere *prims, uint32_t n_prims,
                  const ptor_light *lights, uint32_t n_lights, const float *in, float *out, uint32_t count);

/* ptor_render_with over spheres AND triangles (tris may be NULL when n_tris == 0).  Path integrator
 * only when there are triangles: the reference's light sampling (Primitive::Sample, primitive.h:50-72)
 * is written for spheres.  Returns -1 otherwise. */
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
                    int n_threads);

/* Test hook: one ray against one triangle with the bounds (tmin, tmax): t, or -1. */
float ptor_triangle_intersect(const ptor_triangle *tri, const float o[3], const float d[3], float tmin, float tmax);

#ifdef __cplusplus
}
#endif
#endif /* PT_ORACLE_H */
