/*
 * libptmi -- C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * Drop-in boundary: the five methods of the reference's `class Pathtracer`
 * (pathtracer.h:38-65; implementation pathtracer.cu:172-258) and the scene/camera PODs they
 * consume.  Plain pointers and sizes only; every function returns 0 on success or a negative
 * pt_status, never exits the process (the reference's CudaSafeCall, cutil.h:27-52, calls
 * exit(-1)), and pt_last_error() returns the reason as text.
 *
 * Rendering runs on the current HIP device, on the stream given by pt_set_stream() (default:
 * the null stream).  There is no CPU fallback: without a usable gfx950 device every rendering
 * call fails with PT_ERR_DEVICE.
 *
 * Random numbers: the reference's per-frame cuRAND buffer (pathtracer.cu:207,224) is replaced
 * by Philox4x32-10 keyed by `seed` and addressed by (pixel index, iteration, slot, stream);
 * stream 0 slots 0..1 are a pixel's 8 camera jitters (pathtracer.cu:70), stream 1 slot b holds
 * bounce b's {u, v, roulette} (pathtracer.cu:131,141,155).  uint32 -> float is
 * (x >> 8) * 2^-24.
 */
#ifndef PTMI_H
#define PTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pt_context *pt_handle;

typedef enum {
    PT_OK = 0,
    PT_ERR_INVALID = -1, /* bad argument */
    PT_ERR_STATE = -2,   /* call order (e.g. pt_run before pt_init_scene) */
    PT_ERR_DEVICE = -3,  /* HIP error, or no gfx950 device */
    PT_ERR_NOMEM = -4
} pt_status;

/* What Camera::GenerateRay reads (camera.h:66-72): the members Camera::UpdateParameter
 * derives (camera.h:144-158) plus the image size (camera.h:165-166). */
typedef struct {
    float pos[3];
    float px_x[3];
    float px_y[3];
    float first_ray_dir[3];
    uint32_t width, height;
} pt_camera_pod;

/* material.h:25 */
enum { PT_DIFFUSE = 0, PT_SPECULAR = 1, PT_TRANSMISSIVE = 2 };
/* material.h:45-47 */
typedef struct {
    float color[3];
    float coef;
    int32_t type;
} pt_material;

/* primitive.h:78-81 (the bounding box, primitive.h:77, is never read by the trace loop) */
typedef struct {
    float pos[3];
    float radius;
    uint32_t material_id;
    int32_t light_id; /* -1: not a light (primitive.h:31,75) */
} pt_sphere;

/* A triangle primitive: NO counterpart in the reference, whose only primitive is the sphere
 * ("will be changed to triangle later", primitive.h:25-28); results on scenes with triangles are
 * therefore not pinned to the reference, only to this build's own CPU oracle.  Triangle k has the
 * primitive index n_spheres + k (hit ids, pt_light.u.prim_id).  Intersection: Moeller-Trumbore in
 * fp32, both sides, t within (EPSILON, tmax) like Primitive::Intersect (primitive.h:39-45); the
 * geometric normal normalize((v1 - v0) x (v2 - v0)) stands in for Primitive::Normal and is never
 * flipped (emission is one-sided with respect to it, light.h:43-45). */
typedef struct {
    float v0[3], v1[3], v2[3];
    uint32_t material_id;
    int32_t light_id; /* -1: not a light */
} pt_triangle;

/* light.h:34 */
enum { PT_POINT_LIGHT = 0, PT_AREA_LIGHT = 1 };
/* light.h:47-52 */
typedef struct {
    int32_t type;
    union {
        float pos[3];
        uint32_t prim_id;
    } u;
    float intensity[3];
} pt_light;

/* Frame-buffer update rule of one iteration (1-based number `it`, pixel colour `c`):
 *   PT_ACCUM_RUNNING_MEAN  buffer = (buffer*(it-1) + c) * (1/it)     RenderKernel,
 *                          pathtracer.cu:104-109 (iteration 1 overwrites the buffer)
 *   PT_ACCUM_SUM           buffer += c; the first iteration after pt_reset() overwrites.
 *                          For sharding a render by sample range over several GPUs. */
enum { PT_ACCUM_RUNNING_MEAN = 0, PT_ACCUM_SUM = 1 };

/* hit_ids sentinels of pt_debug_trace */
enum { PT_HIT_MISS = -1, PT_HIT_NOT_TRACED = -2 };

/* ---- life cycle ------------------------------------------------------------------------ */

/* Pathtracer::Pathtracer(Camera*, int nBounces), pathtracer.h:40-43.  The camera is copied;
 * width/height are latched here as in the reference. */
int pt_create(pt_handle *out, const pt_camera_pod *camera, int max_bounces, uint64_t seed);

/* Pathtracer::Init, pathtracer.cu:172-220: uploads the scene (repacked to SoA).  The arrays
 * may be freed after the call.  n_materials is honoured (the reference copies pv.size()
 * materials, pathtracer.cu:194).  A material whose type is none of the three scatters nothing:
 * SampleMaterial leaves f black for it (scene.h:179,220) and the path ends there. */
int pt_init_scene(pt_handle h, const pt_material *materials, uint32_t n_materials,
                  const pt_sphere *spheres, uint32_t n_spheres,
                  const pt_light *lights, uint32_t n_lights);

/* pt_init_scene with triangles beside the spheres (either array may be empty).  An extension without a
 * counterpart in the reference (see pt_triangle).  Scenes with triangles are rendered by the linear scan
 * (spheres, then triangles, in index order: the lowest index wins equal distances, scene.h:78-80) through
 * the path pool -- PT_VARIANT_AUTO, PT_INTEGRATOR_PATH, pt_run / pt_run_n / pt_run_async; no BVH is built
 * for them, and the other kernel variants, pt_run_n_noalloc and the NEE integrator (whose light sampling,
 * primitive.h:50-72, is written for spheres) answer PT_ERR_STATE. */
int pt_init_scene_triangles(pt_handle h, const pt_material *materials, uint32_t n_materials,
                            const pt_sphere *spheres, uint32_t n_spheres,
                            const pt_triangle *triangles, uint32_t n_triangles,
                            const pt_light *lights, uint32_t n_lights);

/* Pathtracer::Release, pathtracer.cu:249-258, plus destruction of the handle. */
int pt_destroy(pt_handle h);

/* Text of the last error of this handle (h may be NULL: last error of pt_create). */
const char *pt_last_error(pt_handle h);

/* ---- rendering ------------------------------------------------------------------------- */

/* Pathtracer::Run, pathtracer.cu:222-243: renders iteration ++iteration (4 jittered paths per
 * pixel) and updates `buffer` (width*height*3 floats, interleaved RGB, row 0 on top).
 * buffer_is_device != 0: a device pointer, updated in place like the reference's mapped PBO.
 * buffer_is_device == 0: a host pointer; the library keeps the device copy and writes the
 * current image to it.  Blocking: returns after the kernel has completed. */
int pt_run(pt_handle h, float *buffer, int buffer_is_device);

/* n_iterations consecutive Run() calls in ONE launch on a device buffer: each pixel keeps its
 * accumulator in registers, results are identical to n calls of pt_run.  sync == 0 returns
 * after enqueueing on the stream. */
int pt_run_n(pt_handle h, float *device_buffer, uint32_t n_iterations, int sync);

/* The reference's frame loop calls Run once per iteration (Display, main.cpp:205-232) and Run
 * blocks until its kernel is done (pathtracer.cu:241), so consecutive iterations can never
 * overlap on the device.  pt_run_async is pt_run_n without the wait, and more: consecutive CALLS
 * alternate between the library's two sets of batch buffers and two streams, the way the batches
 * inside one pt_run_n call do, so the tail of one call's launch (its last, longest paths) runs
 * under the next call's launch.  Results are those of pt_run / pt_run_n, bit for bit: a call's
 * accumulation into the frame buffer waits for the accumulation of the call before it.
 * pt_wait ends the series: it returns when every call so far has completed and the buffer holds
 * their result.  Between pt_run_async and pt_wait the caller must not touch the buffer, also not
 * through its own stream: part of the work runs on a stream of the library's own.  Every other
 * entry point of this handle may be called in between (camera changes, pt_reset, a blocking
 * pt_run, pt_read_image, ...): each of them first orders itself behind the calls made so far.
 * Renders that do not go through the batch buffers (the megakernel variants, the NEE integrator)
 * or do not fit half of them are simply enqueued on the handle's stream. */
int pt_run_async(pt_handle h, float *device_buffer, uint32_t n_iterations);
int pt_wait(pt_handle h);

/* pt_run_n for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same
 * results as pt_run_n, bit for bit; the call itself performs no hipMalloc / hipFree: the scratch
 * of the work-stealing megakernel (blocks for results that change hands between lanes, tickets
 * of the iteration ranges) is a statically allocated 48.3 MiB array that every GPU the library
 * runs on holds until the process ends.  Limits: frames of at most 65 535 tiles of 16x16 pixels
 * (e.g. 4096x4080; PT_ERR_INVALID beyond); megakernel variants only -- PT_VARIANT_AUTO renders
 * BVH scenes through the BVH megakernel here, PT_VARIANT_WAVEFRONT is refused with PT_ERR_STATE
 * (its queues grow with the frame).  All calls on one GPU share that GPU's array: calls that
 * overlap in time on one GPU (different handles or streams) race; calls on one stream do not. */
int pt_run_n_noalloc(pt_handle h, float *device_buffer, uint32_t n_iterations, int sync);

/* Pathtracer::Reset, pathtracer.cu:245-247: iteration = 0. */
int pt_reset(pt_handle h);

/* Pathtracer::GetIteration, pathtracer.h:49. */
uint32_t pt_iteration(pt_handle h);

/* The reference dereferences its Camera* on every Run (pathtracer.cu:232); a C caller
 * passes the new value explicitly.  Width and height must not change. */
int pt_set_camera(pt_handle h, const pt_camera_pod *camera);

/* ---- sharding / plumbing (no counterpart in the single-GPU reference) -------------------- */

int pt_set_stream(pt_handle h, void *hip_stream);
int pt_set_accum_mode(pt_handle h, int mode);
/* Next pt_run renders iteration `iteration`+1: a rank that owns samples [a+1, a+n] calls
 * pt_reset(); pt_set_iteration(a). */
int pt_set_iteration(pt_handle h, uint32_t iteration);
/* Tile split: the image is cut into horizontal bands of band_rows rows (a multiple of 16);
 * this instance renders bands b with b % world == rank and leaves other pixels untouched.
 * world == 1 (default) renders everything. */
int pt_set_row_bands(pt_handle h, uint32_t band_rows, uint32_t rank, uint32_t world);
/* Total number of Intersect calls ("rays", one per path bounce) since creation or the last
 * pt_reset_counters; waits for the stream. */
int pt_ray_count(pt_handle h, uint64_t *out);
int pt_reset_counters(pt_handle h);
/* Kernel variant; results are bit-identical in all of them.
 *   PT_VARIANT_AUTO (default), when the integrator is PT_INTEGRATOR_PATH and the buffers fit the
 *     memory limit (row bands included: the batch then holds the owned rows only): PT_VARIANT_WAVEFRONT for scenes rendered through the
 *     sphere BVH; the path pool for scenes rendered by the linear scan: persistent waves
 *     whose lanes take the next camera ray of the batch, whatever pixel it belongs to, when their
 *     path has ended, 12.25 bytes per path in flight (radiance, stored for the paths that met a
 *     light, and a mask byte per pixel-iteration) folded into the frame in the
 *     reference's order afterwards, batches on two streams like PT_VARIANT_WAVEFRONT's.
 *     Everything else: PT_VARIANT_STEALING
 *   PT_VARIANT_STEALING: PT_VARIANT_REGENERATION, and a lane that has finished its pixel takes
 *     work over from the lane of its wave with the most left: whole pixel-iterations, or single
 *     paths in launches of up to 5 iterations (pt_run's one iteration per call).  Applies to every
 *     megakernel launch (linear scan or BVH, either integrator); allocates a 48 MiB scratch pool
 *     on first use, whatever the frame size (a failed allocation: PT_VARIANT_REGENERATION).
 *     Launches of 32 iterations or more over a small frame are also cut into consecutive
 *     iteration ranges rendered by separate workgroups (up to 8, at least 16 iterations each), so
 *     that fewer SIMDs idle while the last workgroups finish
 *   PT_VARIANT_REGENERATION: megakernel, persistent lanes with path regeneration
 *   PT_VARIANT_CLASSIC: megakernel, a wave starts new paths only when all its lanes are idle
 *   PT_VARIANT_WAVEFRONT: one launch per bounce over compacted path queues in device memory
 *     (100 bytes per path in flight, or 56 / 12 for BVH scenes whose persistent waves follow whole
 *     paths, see PT_TUNE_WAVEFRONT_WHOLE_PATHS, and a mask byte per four paths, planned as one per
 *     path; up to about 6.7 GB, allocated on first use -- see
 *     pt_memory_footprint and pt_set_memory_limit).  A render of more iterations
 *     than fit runs as batches that alternate between two buffer sets on two streams (the
 *     caller's and one of the instance's), so that one batch's launches fill the chip while the
 *     other's last traversals finish; the caller's stream ends behind both.  BVH scenes:
 *     persistent waves that refill idle lanes from the queue while the other lanes keep traversing
 * Selecting a variant resets the tuning knobs below to their defaults. */
enum { PT_VARIANT_AUTO = -1, PT_VARIANT_REGENERATION = 0, PT_VARIANT_CLASSIC = 1, PT_VARIANT_WAVEFRONT = 100,
       PT_VARIANT_STEALING = 500 };
int pt_set_kernel_variant(pt_handle h, int variant);
/* Tuning knobs of the kernels, for measurements and tests (results stay bit-identical); not
 * part of the drop-in surface.  Call after pt_set_kernel_variant.
 *   PT_TUNE_REGEN_MIN_IDLE     1..64: idle lanes a wave waits for before it starts new paths
 *                              (1: PT_VARIANT_REGENERATION's default, 64: PT_VARIANT_CLASSIC)
 *   PT_TUNE_CAMERA_RAY_RING    0/1: per-lane ring of pre-generated camera rays in LDS (scenes of
 *                              up to 24 spheres; default 1)
 *   PT_TUNE_STEAL_UNIT         0: chosen per launch; 1 / 4: single paths / whole pixel-iterations
 *                              change hands
 *   PT_TUNE_ITERATION_RANGES   0: chosen per launch; n: a launch is cut into n iteration ranges
 *   PT_TUNE_WAVEFRONT_REFILL   BVH scenes: lanes still traversing below which a persistent wave
 *                              refills (default 32); 0: static assignment of paths to lanes
 *   PT_TUNE_BVH_NODE_STEPS     1..16: inner-node steps per leaf step of the BVH traversal (default 2;
 *                              reset by pt_init_scene, not by pt_set_kernel_variant)
 *   PT_TUNE_BVH_STACK          entries of the per-lane traversal stack (0: the default, two per
 *                              tree level + 2; a ray that runs out is finished by the wave-wide
 *                              scan; reset by pt_init_scene)
 *   PT_TUNE_WAVEFRONT_BATCH    paths in flight of the wavefront variant in units of 2^20 (0: the
 *                              default; kept across variant changes)
 *   PT_TUNE_POOL_ITERATIONS    PT_VARIANT_AUTO, linear-scan scenes: launches of up to this many
 *                              iterations go through the path pool (default 2^30; 0: never, the
 *                              work-stealing megakernel renders; kept across variant changes)
 *   PT_TUNE_WAVEFRONT_WHOLE_PATHS  BVH scenes, how bounces map to launches: 0: one launch per
 *                              bounce over two queues (100 bytes per path in flight); 2: the
 *                              camera rays in a launch of their own, then one launch that follows
 *                              every surviving path to its end (one queue, 56 bytes; a third of the
 *                              memory traffic, 0.7 % slower on the 10 007-sphere scene); 1: one
 *                              launch follows every path from its camera ray on (no queue, 12
 *                              bytes; 5 % slower); -1 (default): 2, or 1 when the memory limit has no
 *                              room for one iteration's queue (kept across variant changes)
 *   PT_TUNE_BUFFER_SETS        path pool and wavefront launches: 0 (default): two sets of batch buffers,
 *                              consecutive batches alternate between them on two streams and overlap;
 *                              1: one set on the caller's stream, every launch after the one before --
 *                              slower, but a kernel trace of it is additive (profiling; kept across
 *                              variant changes)
 *   PT_TUNE_POOL_WORKGROUPS    path pool: persistent workgroups per CU of a launch, 1..8 (0: the default --
 *                              as many as fit, at most 8; calls of a pt_run_async series take fewer, so that
 *                              two consecutive calls' launches are resident side by side; kept across
 *                              variant changes) */
enum { PT_TUNE_REGEN_MIN_IDLE = 0, PT_TUNE_CAMERA_RAY_RING = 1, PT_TUNE_STEAL_UNIT = 2, PT_TUNE_ITERATION_RANGES = 3,
       PT_TUNE_WAVEFRONT_REFILL = 4, PT_TUNE_BVH_NODE_STEPS = 5, PT_TUNE_BVH_STACK = 6, PT_TUNE_WAVEFRONT_BATCH = 7,
       PT_TUNE_WAVEFRONT_WHOLE_PATHS = 8, PT_TUNE_POOL_ITERATIONS = 9, PT_TUNE_BUFFER_SETS = 10, PT_TUNE_POOL_WORKGROUPS = 11 };
int pt_debug_set_tuning(pt_handle h, int key, int value);
/* Device memory held by the instance: bytes[0] = allocated now (scene, BVH, scratch pool,
 * tickets, wavefront queues, internal frame buffer); bytes[1] = what a pt_run_n of n_iterations
 * would hold at its peak with the current settings.  The statically allocated scratch of
 * pt_run_n_noalloc (48.3 MiB per GPU, per process) is not included. */
int pt_memory_footprint(pt_handle h, uint32_t n_iterations, uint64_t bytes[2]);
/* Upper bound, in bytes, on what the instance allocates for scratch and queues (0: none, the
 * default).  The wavefront variant sizes its batches to fit -- or, under PT_VARIANT_AUTO, leaves
 * the render to the megakernel when not even one iteration's queues fit (PT_VARIANT_WAVEFRONT
 * itself then fails with PT_ERR_NOMEM); a limit below 48 MiB also switches work stealing off. */
int pt_set_memory_limit(pt_handle h, uint64_t bytes);
/* Integrator.
 *   PT_INTEGRATOR_PATH (default): Trace as the reference runs it, pathtracer.cu:112-170.
 *   PT_INTEGRATOR_NEE_MIS: adds direct-light sampling with multiple importance sampling, wiring
 *     the functions the reference carries but never calls -- SampleLight scene.h:150-170,
 *     Pdf scene.h:110-144, Primitive::Sample primitive.h:50-72, PowerHeuristic
 *     montecarlo.h:156-159 -- and the commented-out `bounces == 0 || specularBounce` logic of
 *     pathtracer.cu:117,133,148.  Three of those functions are repaired (primitive.h:67,
 *     scene.h:117-121, scene.h:167) and area-light occlusion is decided by the nearest hit
 *     instead of VisibilityTester's shortened segment; same expectation as PATH, lower variance
 *     where the light is small.  pt_ray_count then includes the shadow rays.  Megakernel
 *     variants only. */
enum { PT_INTEGRATOR_PATH = 0, PT_INTEGRATOR_NEE_MIS = 1 };
int pt_set_integrator(pt_handle h, int integrator);
/* Nearest-hit search of scenes with more than 192 spheres; results are bit-identical.
 *   PT_ACCEL_AUTO (default): sphere BVH built by pt_init_scene -- the acceleration structure
 *     the reference leaves as "TODO kd-tree" (scene.h:33) -- when the scene allows one
 *   PT_ACCEL_BRUTE_FORCE: the reference's linear scan over all spheres (scene.h:71-94),
 *     (from LDS; streamed through it in tiles above 512 spheres)
 * pt_accel_nodes returns the number of BVH nodes the scene got (0: none, the scan is used). */
enum { PT_ACCEL_AUTO = 0, PT_ACCEL_BRUTE_FORCE = 1 };
int pt_set_accel(pt_handle h, int mode);
uint32_t pt_accel_nodes(pt_handle h);
/* A ray that has visited more than `visits` boxes and leaves stops descending and is finished
 * by its whole wave scanning all spheres (same result; bounds the cost of the few rays whose
 * boxes must be widened enormously).  0 restores the default, max(64, spheres / 40), set by
 * pt_init_scene.  A tuning and test knob. */
int pt_set_accel_budget(pt_handle h, uint32_t visits);
/* Test hook, host only (no device needed): builds the BVH pt_init_scene would build for these
 * spheres, whatever their number.  counts = {inner nodes, leaves, always-tested spheres, depth}
 * on return, all 0 when no hierarchy is built.  The arrays (may be NULL) are filled only when
 * counts holds the first three values on entry: nodes [inner nodes][32] floats (a four-wide node
 * as laid out in device memory: lo.x, lo.y, lo.z, hi.x, hi.y, hi.z of the four children, their
 * four references as int bits -- >= 0 inner node, < 0 ~leaf, 0x7fffffff unused slot -- and four
 * unused floats), leaf_index [leaves][4] sphere indices (0x7fffffff = empty slot), big_index
 * [always-tested].  Call once to size, once to fill. */
int pt_debug_build_bvh(const pt_sphere *spheres, uint32_t n_spheres, uint32_t counts[4], float *nodes,
                       int32_t *leaf_index, int32_t *big_index);
/* Test hook, host only: the 16-bit grid form of the same nodes, which is what the device reads.
 * grid = {origin x, y, z, step x, y, z} (plane = origin + q * step); qnodes [n_nodes][16] words of a
 * node with n_nodes = counts[0] of pt_debug_build_bvh: words 0..11 hold the planes lo.x, lo.y, lo.z,
 * hi.x, hi.y, hi.z, two words each (children 0|1 and 2|3, the lower child in the low half), words
 * 12..15 the four child references.  PT_ERR_STATE when no hierarchy is built for these spheres. */
int pt_debug_bvh_grid(const pt_sphere *spheres, uint32_t n_spheres, float grid[6], uint32_t *qnodes, uint32_t n_nodes);
/* Test hook, host only (no device needed): how a render of n_iterations over a width x height
 * frame is cut into batches by the path pool and the wavefront variant, with row bands
 * (band_world <= 1: none) and bytes_per_path of buffers per path in flight (the library plans
 * with 13, 57 or 101) under
 * memory_limit (0: the default budget).  out = {rows this rank owns, iterations per batch (0: not
 * one iteration fits), buffer sets (1 or 2; batches alternate between them), slots per
 * sub-queue}. */
int pt_debug_plan_batches(uint32_t width, uint32_t height, uint32_t band_rows, uint32_t band_rank, uint32_t band_world,
                          uint32_t n_iterations, uint32_t bytes_per_path, uint64_t memory_limit, uint32_t out[4]);
/* Traversal counters since the last call (only libraries built with -DPT_BVH_STATS count -- the
 * counting costs time -- others return zeros):
 *   [0] inner nodes visited  [1] leaves visited  [2] rays traversed  [3] rays finished by the wave-wide scan
 *   [4] inner-node iterations and [5] leaf iterations summed over waves (a wave iterates until
 *   its slowest lane is done)  [6] wave-level traversals  [7] rays with a direction longer than 1
 *   [8 + k] rays whose visit count n satisfies 2^k <= n + 1 < 2^(k+1) */
int pt_accel_stats(pt_handle h, uint64_t out[32]);
/* Divergence counters of the megakernel since the last call (only libraries built with
 * -DPT_LANE_STATS count, others return zeros): for code site i, out[2*i] = times a wave executed
 * the site, out[2*i+1] = lanes active in those executions.  Sites: 0 camera-ray ring refill,
 * 1 path start, 2 sphere test, 3 its square-root part, 4 loop pass, 5 bounce shading, 6 emission,
 * 7 throughput update, 8 Russian roulette, 10 BSDF sample, 11/13 its glass-only parts, 12/14 its
 * diffuse-only parts, 15 mirror direction, 16 normalised direction, 17 lanes with a ray in a pass,
 * 18 path end, 19 pixel-iteration end, 20 steal attempt, 21 steal, 22 BVH box re-test with the
 * narrowed widening (two per inner step at most), 23 BVH inner step, 24 BVH leaf step.  Has no counterpart in the reference (measurement only). */
int pt_lane_stats(pt_handle h, uint64_t out[64]);
/* HIP-event time of the last pt_run/pt_run_n launch in milliseconds (waits for it). */
int pt_last_kernel_ms(pt_handle h, float *ms);
/* Which kernels the last pt_run / pt_run_n / pt_run_n_noalloc went through (what PT_VARIANT_AUTO
 * resolved to for that scene and launch).  Has no counterpart in the reference (measurement only). */
enum { PT_LAUNCH_NONE = 0, PT_LAUNCH_MEGAKERNEL = 1, PT_LAUNCH_MEGAKERNEL_STEALING = 2, PT_LAUNCH_WAVEFRONT = 3,
       PT_LAUNCH_WAVEFRONT_WHOLE_PATHS = 4, PT_LAUNCH_PATH_POOL = 5 };
int pt_last_launch(pt_handle h, int *kind);

/* Copies a device frame buffer to the host (pt_read_image(h, dev, host)); with PT_ACCUM_SUM
 * the caller divides by the number of iterations. */
int pt_read_image(pt_handle h, const float *device_buffer, float *host_rgb);

/* Checkpoint / resume (no counterpart: the reference's state is `iteration` and the frame
 * buffer, pathtracer.h:58, and is lost with the window).  The file holds the frame buffer, the
 * iteration counter, seed, depth, size and accumulation mode; loading into an instance created
 * with different values fails with PT_ERR_STATE.  Scene and camera are the caller's to restore.
 * A render resumed from a checkpoint is bit-identical to an uninterrupted one. */
int pt_save_checkpoint(pt_handle h, const float *device_buffer, const char *path);
int pt_load_checkpoint(pt_handle h, float *device_buffer, const char *path);

/* Parity hook: traces iteration `iteration` of the pixel rectangle [x0,x1) x [y0,y1) with the
 * production device functions and returns, per path ([pixel row-major][sub-sample 0..3]):
 *   hit_ids  [..][max_bounces+1] int32  primitive index of the k-th Intersect call
 *            (scene.h:71-94), PT_HIT_MISS, or PT_HIT_NOT_TRACED
 *   radiance [..][3] float              the Color Trace() returned (pathtracer.cu:169)
 *   pixel_rgb[pixel][3] float           (L0+L1+L2+L3)*(1/4), pathtracer.cu:94-101
 * Host pointers, any may be NULL.  Does not touch the frame buffer or the iteration. */
int pt_debug_trace(pt_handle h, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                   uint32_t iteration, int32_t *hit_ids, float *radiance, float *pixel_rgb,
                   uint64_t *ray_count);

/* ---- arithmetic self-tests (no counterpart in the reference) --------------------------------
 * The trace loop computes sqrt, 1/x and a/b with short correctly rounded sequences and sin/cos
 * with a fixed fp32 polynomial so that the device matches a host evaluation bit for bit.
 * pt_selftest_eval: out[i] = op(a[i], b[i]) on the device; op 0 sqrt(a), 1 1/a, 2 a/b,
 *   3 sin(a), 4 cos(a) (b may be NULL unless op == 2).  Host pointers.
 * pt_selftest_exhaustive: number of results that differ from the compiler's generic IEEE
 *   expansion over all 2^32 float bit patterns (op 0, 1; op 3: the square root of the sphere
 *   test's discriminant, which may return NaN for +infinity) or 2^32 random operand pairs (op 2). */
int pt_selftest_eval(pt_handle h, int op, const float *a, const float *b, float *out, uint64_t n);
int pt_selftest_exhaustive(pt_handle h, int op, uint64_t *mismatches);
/* SampleMaterial (scene.h:177-221) as the kernels evaluate it, row by row on the device: in
 * [n][8] = wo.xyz, n.xyz, u, v; out [n][8] = f.rgb, wi.xyz, pdf, 0.  Host pointers.  For a material
 * of unknown type only f (black) is defined. */
int pt_selftest_sample_material(pt_handle h, const pt_material *material, const float *in, float *out,
                                uint64_t n);

/* ---- host-side scene helpers (plain host code, no device needed) ------------------------- */

/* Camera::Camera + UpdateParameter, camera.h:31-57,144-158. */
int pt_camera_init(pt_camera_pod *out, const float eye[3], const float look_at[3],
                   const float up[3], uint32_t width, uint32_t height, float fov_deg);

/* Interactive camera (camera.h:79-129,144-158): the full state Translate/Rotate work on.  `pod` is
 * kept up to date and is what pt_create / pt_set_camera take. */
typedef struct {
    float view[3][3];   /* rows u, v, w of the view matrix (camera.h:49-54) */
    float dist, hfov, vfov;
    pt_camera_pod pod;
} pt_camera_state;

/* Camera::Camera, camera.h:31-57 (same result in `pod` as pt_camera_init). */
int pt_camera_state_init(pt_camera_state *cam, const float eye[3], const float look_at[3],
                         const float up[3], uint32_t width, uint32_t height, float fov_deg);
/* Camera::Translate, camera.h:79-90: pos += u*dx + v*dy + w*dz. */
int pt_camera_translate(pt_camera_state *cam, const float delta[3]);
/* Camera::Rotate, camera.h:97-129 (theta.z is ignored as in the reference), then UpdateParameter. */
int pt_camera_rotate(pt_camera_state *cam, const float theta[3]);

typedef enum {
    PT_SCENE_DEMO = 0,      /* InitScene "scene 2", main.cpp:129-164: 9 spheres, 1 area light */
    PT_SCENE_SYNTH10K = 1,  /* demo room + 10 000 procedural spheres */
    PT_SCENE_MIXED = 2      /* demo room + 8x8 grid cycling diffuse/specular/glass */
} pt_scene_id;

/* Two-call pattern: with NULL arrays only the counts are written.  eye/look_at/up/fov_deg
 * (may be NULL) receive the camera of main.cpp:129. */
int pt_scene_build(int scene_id, uint64_t seed,
                   pt_material *materials, uint32_t *n_materials,
                   pt_sphere *spheres, uint32_t *n_spheres,
                   pt_light *lights, uint32_t *n_lights,
                   float eye[3], float look_at[3], float up[3], float *fov_deg);

/* Library/ABI version: major*10000 + minor*100 + patch. */
int pt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PTMI_H */
