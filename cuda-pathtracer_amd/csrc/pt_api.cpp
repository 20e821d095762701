// C ABI of libptmi (include/ptmi.h): host-side driver of the render kernels.
//
// Mirrors the life cycle of the reference's Pathtracer (pathtracer.h:38-65,
// pathtracer.cu:172-258): create -> Init (scene upload) -> Run per iteration -> Reset /
// Release.  There is deliberately no CPU rendering path: every rendering entry point needs a
// gfx950 device and reports PT_ERR_DEVICE otherwise.
#include "ptmi.h"
#include "pt_kernels.hpp"
#include "pt_bvh.hpp"
#include "pt_checkpoint.hpp"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace ptmi;

#ifdef PT_SORT_EXPERIMENT
namespace ptmi {
hipError_t sort_queue_experiment(const WavefrontParams &p, WavefrontQueue from, WavefrontQueue to, const uint32_t *counts, int mode,
                                 hipStream_t stream);
}
#include <cstdlib>
#endif

struct pt_context {
    pt_camera_pod camera;
    int max_bounces;
    uint64_t seed;
    uint32_t iteration;   // Pathtracer::iteration, pathtracer.h:58
    bool fresh;           // nothing accumulated since create/reset
    int accum_mode;
    uint32_t band_rows, band_rank, band_world;
    int regen_min_idle;
    int small_scene_opts;
    hipStream_t stream;
    int device;
    // scene (device)
    float4 *d_sph, *d_aux, *d_mat, *d_light, *d_light_geom;
    float4 *d_tri;        // triangles, four float4 each (pt_init_scene_triangles); NULL when the scene has none
    uint32_t n_tris;
    int2 *d_mat_type;   // (type, bits of 1.f / coef)
    int integrator;       // PT_INTEGRATOR_*
    uint32_t n_prims, n_mats, n_lights;
    bool scene_ready;
    uint64_t scene_hash;  // fingerprint of the arrays given to pt_init_scene (checkpoints)
    uint64_t scene_bytes; // device memory of the scene arrays and the BVH
    // sphere BVH of scenes too large for LDS residency (device); n_nodes == 0: none
    BvhParams bvh;
    void *d_bvh;          // one allocation behind the pointers in `bvh`
    uint32_t bvh_default_budget;
    uint32_t bvh_default_stack;   // stack entries upload_bvh chose
    int accel;            // PT_ACCEL_*
    // misc device state
    unsigned long long *d_ray_counter;
    void *d_steal;             // scratch pool of the work-stealing megakernel + its occupancy bits, allocated on first use
    bool steal_alloc_failed;   // the pool could not be allocated: render without stealing, do not try again
    StealScratch fixed_scratch; // this device's copy of the statically allocated scratch (pt_run_n_noalloc)
    bool steal;                // work stealing inside the wave where the kernel supports it
    uint64_t memory_limit;     // pt_set_memory_limit: bytes of device memory the library may allocate; 0 = no limit
    uint32_t *d_tickets;       // chunked launches: start-order counter + one progress word per tile
    size_t tickets_capacity;   // in words
    int chunks;                // 0: chosen per launch; n: iteration ranges per launch (experiments)
    int steal_unit;            // 0: chosen per launch; 1 or 4: paths per stolen unit (experiments)
    float *d_internal_buffer; // frame buffer owned by the library when the caller passes host memory
    hipEvent_t ev_start, ev_stop;
    bool timed;
    // wavefront variant: queues, per-path radiance and counters, allocated on first use
    bool wavefront;
    bool auto_variant;
    int wf_dynamic_fetch;   // BVH scenes: refill threshold of the persistent waves, 0 = static chunks
    // Two sets of batch buffers: consecutive batches alternate between them and run on two streams
    // (set 0 on the caller's stream, set 1 on wf_stream2), so that the tail of one batch's launch --
    // the last, longest traversals -- overlaps with the other batch's launches.
    struct WfPipe {
        WavefrontQueue queue[2];
        float *radiance;
        uint32_t *counts;
        hipEvent_t resolved;    // recorded after the set's resolve kernel
        bool counts_dirty;      // the counters are not known to be zero (fresh allocation, failed batch)
    } wf_pipe[2];
    hipStream_t wf_stream2;
    hipEvent_t wf_fork;
    // pt_run_async: consecutive CALLS alternate between the two buffer sets and streams the way the
    // batches of one call do.  async_series: a series of such calls is open (wf_fork marks where the
    // caller's stream stood when it began); async_set: the set the next call takes; async_on_stream2:
    // the series' last call went to the second stream, which the caller's stream has not waited for yet.
    bool async_series, async_on_stream2, async_forked;
    int async_set;
    uint32_t wf_target_mpaths;   // paths per wavefront batch in units of 2^20 (0: kWfTargetPaths)
    int wf_whole_paths;          // BVH scenes with dynamic fetch, PT_TUNE_WAVEFRONT_WHOLE_PATHS: -1 chosen per render
    int wf_n_queues;             // path queues per buffer set (0, 1 or 2)
    int pool_workgroups;         // PT_TUNE_POOL_WORKGROUPS: persistent workgroups per CU of a pool launch, 0 = default
    int wf_buffer_sets;          // PT_TUNE_BUFFER_SETS: 0 two sets on two streams, 1 one set on the caller's stream
    int pool_max_iterations;     // resident scenes: launches of up to this many iterations go through the path pool
    int last_launch;             // PT_LAUNCH_* of the last render
    uint32_t wf_sub_capacity;   // paths per sub-queue the buffers were sized for
    uint64_t wf_capacity;       // paths per batch the buffers were sized for
    uint64_t wf_mask_bytes;     // size of the radiance mask behind a set's counters
    std::string error;
};

static thread_local std::string g_create_error;

static int fail(pt_context *h, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->error = buf; else g_create_error = buf;
    return code;
}

#define PT_HIP(h, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((h), PT_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e_));    \
    } while (0)

// An instance belongs to the device that was current at pt_create (pt_context::device).  Every
// entry point that reaches the HIP runtime makes that device current for the duration of the
// call, so a process that drives several devices may call in with any device current.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// Ends a series of pt_run_async calls: the caller's stream gets behind everything the series put
// on the second stream.  The resolves of consecutive calls wait for each other, so the last call's
// resolve is behind all of them: when it ran on the caller's stream nothing is left to wait for.
// Every entry point that uses the caller's stream (or frees what the launches use) calls this first.
static hipError_t join_async(pt_context *h)
{
    if (h->async_on_stream2) {
        const hipError_t e = hipStreamWaitEvent(h->stream, h->wf_pipe[1].resolved, 0);
        if (e != hipSuccess) return e;
    }
    h->async_series = h->async_on_stream2 = h->async_forked = false;
    h->async_set = 0;
    return hipSuccess;
}

// Scenes are staged whole in LDS (spheres, materials, lights) when they are small enough;
// otherwise spheres stream through an LDS tile and the rest is read from device memory.
static bool scene_streams(const SceneParams &s)
{
    return s.n_prims > kMaxResidentSpheres || s.n_tris > kMaxResidentTriangles || scene_lds_bytes(s, false) > kMaxResidentBytes;
}

static bool camera_ok(const pt_camera_pod *c)
{
    return c && c->width > 0 && c->height > 0 && (uint64_t)c->width * c->height <= 0x7fffffffull / 3;
}

extern "C" {

int pt_version(void) { return 100; }

const char *pt_last_error(pt_handle h) { return h ? h->error.c_str() : g_create_error.c_str(); }

int pt_create(pt_handle *out, const pt_camera_pod *camera, int max_bounces, uint64_t seed)
{
    if (!out) return fail(nullptr, PT_ERR_INVALID, "pt_create: out is NULL");
    *out = nullptr;
    if (!camera_ok(camera)) return fail(nullptr, PT_ERR_INVALID, "pt_create: bad camera (width/height)");
    if (max_bounces < 0 || max_bounces > 1024)
        return fail(nullptr, PT_ERR_INVALID, "pt_create: max_bounces %d out of range [0,1024]", max_bounces);

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0)
        return fail(nullptr, PT_ERR_DEVICE, "pt_create: no HIP device (%s); libptmi has no CPU path",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    int dev = 0;
    PT_HIP(nullptr, hipGetDevice(&dev));
    hipDeviceProp_t prop;
    PT_HIP(nullptr, hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PT_ERR_DEVICE, "pt_create: device %d is %s; libptmi is built for gfx950 only",
                    dev, prop.gcnArchName);

    pt_context *h = new (std::nothrow) pt_context();
    if (!h) return fail(nullptr, PT_ERR_NOMEM, "pt_create: out of host memory");
    h->camera = *camera;
    h->max_bounces = max_bounces;
    h->seed = seed;
    h->iteration = 0;
    h->fresh = true;
    h->accum_mode = PT_ACCUM_RUNNING_MEAN;
    h->band_rows = 0; h->band_rank = 0; h->band_world = 1;
    h->regen_min_idle = 1;
    h->small_scene_opts = 1;
    h->stream = nullptr;
    h->device = dev;
    h->d_sph = h->d_aux = h->d_mat = h->d_light = h->d_light_geom = nullptr;
    h->d_mat_type = nullptr;
    h->d_tri = nullptr;
    h->n_tris = 0;
    h->integrator = PT_INTEGRATOR_PATH;
    h->n_prims = h->n_mats = h->n_lights = 0;
    h->scene_ready = false;
    h->scene_hash = 0;
    h->scene_bytes = 0;
    memset(&h->bvh, 0, sizeof h->bvh);
    h->d_bvh = nullptr;
    h->accel = PT_ACCEL_AUTO;
    h->d_ray_counter = nullptr;
    h->d_steal = nullptr;
    h->steal_alloc_failed = false;
    memset(&h->fixed_scratch, 0, sizeof h->fixed_scratch);
    h->steal = true;           // PT_VARIANT_AUTO
    h->memory_limit = 0;
    h->d_tickets = nullptr;
    h->tickets_capacity = 0;
    h->chunks = 0;
    h->steal_unit = 0;
    h->d_internal_buffer = nullptr;
    h->ev_start = h->ev_stop = nullptr;
    h->timed = false;
    h->wavefront = false;
    h->auto_variant = true;
    h->wf_dynamic_fetch = 32;
    memset(h->wf_pipe, 0, sizeof h->wf_pipe);
    h->wf_stream2 = nullptr;
    h->wf_fork = nullptr;
    h->async_series = h->async_on_stream2 = h->async_forked = false;
    h->async_set = 0;
    h->wf_target_mpaths = 0;
    h->wf_whole_paths = -1;
    h->wf_n_queues = 0;
    h->wf_buffer_sets = 0;
    h->pool_workgroups = 0;
    h->pool_max_iterations = 1 << 30;
    h->last_launch = PT_LAUNCH_NONE;
    h->wf_sub_capacity = 0;
    h->wf_capacity = 0;
    h->wf_mask_bytes = 0;

    if (hipMalloc((void **)&h->d_ray_counter, sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(h->d_ray_counter, 0, sizeof(unsigned long long)) != hipSuccess ||
        hipEventCreate(&h->ev_start) != hipSuccess || hipEventCreate(&h->ev_stop) != hipSuccess) {
        pt_destroy(h);
        return fail(nullptr, PT_ERR_DEVICE, "pt_create: device allocation failed");
    }
    // address of this device's copy of the statically allocated scratch, looked up once per device
    if (hipError_t se = static_steal_scratch(dev, &h->fixed_scratch); se != hipSuccess) {
        pt_destroy(h);
        return fail(nullptr, PT_ERR_DEVICE, "pt_create: hipGetSymbolAddress of the static scratch failed: %s",
                    hipGetErrorString(se));
    }
    *out = h;
    return PT_OK;
}

static void free_scene(pt_context *h)
{
    if (h->d_sph) (void)hipFree(h->d_sph);
    if (h->d_aux) (void)hipFree(h->d_aux);
    if (h->d_mat) (void)hipFree(h->d_mat);
    if (h->d_light) (void)hipFree(h->d_light);
    if (h->d_mat_type) (void)hipFree(h->d_mat_type);
    if (h->d_light_geom) (void)hipFree(h->d_light_geom);
    if (h->d_tri) (void)hipFree(h->d_tri);
    h->d_tri = nullptr;
    h->n_tris = 0;
    h->d_sph = h->d_aux = h->d_mat = h->d_light = h->d_light_geom = nullptr;
    h->d_mat_type = nullptr;
    if (h->d_bvh) (void)hipFree(h->d_bvh);
    h->d_bvh = nullptr;
    memset(&h->bvh, 0, sizeof h->bvh);
    h->scene_ready = false;
    h->scene_bytes = 0;
}

// Build the sphere BVH on the host and upload it as one allocation.
static int upload_bvh(pt_context *h, const std::vector<float4> &sph, const pt_sphere *spheres, uint32_t n)
{
    std::vector<float> radius(n);
    for (uint32_t i = 0; i < n; ++i) radius[i] = spheres[i].radius;
    BvhData d;
    if (!build_bvh(sph.data(), radius.data(), n, &d)) return PT_OK;   // brute force stays
    auto pad = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
    const size_t n_nodes = d.nodes.size() / kBvhNodeVec4, n_slots = d.leaf_sph.size(), n_big = d.big_sph.size();
    const size_t off_nodes = 0;
    const size_t off_ls = off_nodes + pad(d.qnodes.size() * sizeof(uint4));
    const size_t off_li = off_ls + pad(n_slots * sizeof(float4));
    const size_t off_bs = off_li + pad(n_slots * sizeof(int));
    const size_t off_bi = off_bs + pad((n_big ? n_big : 1) * sizeof(float4));
    const size_t off_st = off_bi + pad((n_big ? n_big : 1) * sizeof(int));
    const size_t total = off_st + 256;
    std::vector<unsigned char> blob(total, 0);
    memcpy(&blob[off_nodes], d.qnodes.data(), d.qnodes.size() * sizeof(uint4));
    memcpy(&blob[off_ls], d.leaf_sph.data(), n_slots * sizeof(float4));
    memcpy(&blob[off_li], d.leaf_index.data(), n_slots * sizeof(int));
    if (n_big) {
        memcpy(&blob[off_bs], d.big_sph.data(), n_big * sizeof(float4));
        memcpy(&blob[off_bi], d.big_index.data(), n_big * sizeof(int));
    }
    PT_HIP(h, hipMalloc(&h->d_bvh, total));
    PT_HIP(h, hipMemcpy(h->d_bvh, blob.data(), total, hipMemcpyHostToDevice));
    h->scene_bytes += total;
    const unsigned char *base = (const unsigned char *)h->d_bvh;
    h->bvh.nodes = (const uint4 *)(base + off_nodes);
    h->bvh.leaf_sph = (const float4 *)(base + off_ls);
    h->bvh.leaf_index = (const int4 *)(base + off_li);
    h->bvh.big_sph = (const float4 *)(base + off_bs);
    h->bvh.big_index = (const int *)(base + off_bi);
    h->bvh.stats = (unsigned long long *)((unsigned char *)h->d_bvh + off_st);
    h->bvh.n_nodes = (uint32_t)n_nodes;
    h->bvh.n_big = (uint32_t)n_big;
    h->bvh.n_leaves = (uint32_t)(n_slots / kBvhLeafSize);
    // Entries of the per-lane traversal stack (LDS, 2 KiB per entry and workgroup).  A node defers up
    // to three children, but a ray rarely meets more than two of four boxes: 2 per level + 2, at
    // most 12 (24 KiB: six workgroups per CU), is ample -- measured on the 10 007-sphere scene: no
    // difference between 10 and 20 entries, 8 costs 15 %.  A ray that does run out gives up the
    // tree and is finished by the wave-wide scan.
    h->bvh.depth = std::min(std::min(2 * d.depth + 2, 3 * d.depth), 12u);
    // A wave-wide scan of n spheres costs about as much as n / 40 divergent visits of four-wide
    // nodes (10 007 spheres, ms per 16 iterations: budget 64 101, 90 93.9, 125 92.6, 180 91.4, 250 91.0, 400 91.4).
    h->bvh_default_budget = n / 40 > 64 ? n / 40 : 64;
    h->bvh.budget = h->bvh_default_budget;
    h->bvh.node_steps = 2;   // measured on the 10 007-sphere scene: 1: 70.5, 2: 68.8, 3: 70.6, 4: 74.4 ms per 8 iterations
    h->bvh_default_stack = h->bvh.depth;
    for (int a = 0; a < 3; ++a) { h->bvh.centre[a] = d.centre[a]; h->bvh.grid_origin[a] = d.grid_origin[a]; h->bvh.grid_step[a] = d.grid_step[a]; }
    h->bvh.half_diag = d.half_diag;
    h->bvh.r_min = d.r_min;
    return PT_OK;
}

static void free_wavefront(pt_context *h)
{
    if (h->wf_stream2) (void)hipStreamSynchronize(h->wf_stream2);
    // nothing is left on the second stream: an open series of pt_run_async calls has nothing to join
    h->async_series = h->async_on_stream2 = h->async_forked = false;
    h->async_set = 0;
    for (auto &pipe : h->wf_pipe) {
        for (auto &q : pipe.queue) {
            if (q.a) (void)hipFree(q.a);
            if (q.b) (void)hipFree(q.b);
            if (q.c) (void)hipFree(q.c);
        }
        if (pipe.radiance) (void)hipFree(pipe.radiance);
        if (pipe.counts) (void)hipFree(pipe.counts);
        if (pipe.resolved) (void)hipEventDestroy(pipe.resolved);
    }
    memset(h->wf_pipe, 0, sizeof h->wf_pipe);
    h->wf_sub_capacity = 0; h->wf_capacity = 0; h->wf_mask_bytes = 0;
    h->wf_n_queues = 0;
}

int pt_destroy(pt_handle h)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (h->async_series) (void)join_async(h);
    (void)hipStreamSynchronize(h->stream);
    free_scene(h);
    free_wavefront(h);
    if (h->d_ray_counter) (void)hipFree(h->d_ray_counter);
    if (h->d_steal) (void)hipFree(h->d_steal);
    if (h->d_tickets) (void)hipFree(h->d_tickets);
    if (h->d_internal_buffer) (void)hipFree(h->d_internal_buffer);
    if (h->ev_start) (void)hipEventDestroy(h->ev_start);
    if (h->ev_stop) (void)hipEventDestroy(h->ev_stop);
    if (h->wf_fork) (void)hipEventDestroy(h->wf_fork);
    if (h->wf_stream2) (void)hipStreamDestroy(h->wf_stream2);
    delete h;
    return PT_OK;
}

// Device table entry of a material: (f, coef) with f the value SampleMaterial returns for it, which
// depends on the material alone -- diffuse: Material::F, 0 + color * INV_PI (material.h:37-43,
// scene.h:186); specular: coef * color (scene.h:191); refractive: color (scene.h:216); any other
// type: black (scene.h:179,220).  Same fp32 operations as the reference performs per bounce
// (this file is compiled without contraction).
static float4 material_table_entry(const pt_material &m)
{
    const float inv_pi = 0.31830988618379067154f;
    float f[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; ++k) {
        if (m.type == 0) f[k] = 0.f + m.color[k] * inv_pi;
        else if (m.type == 1) f[k] = m.color[k] * m.coef;
        else if (m.type == 2) f[k] = m.color[k];
    }
    return make_float4(f[0], f[1], f[2], m.coef);
}

// Second table entry of a material: its type and n1/n2 of a ray that enters a refractive material,
// 1.f / coef (scene.h:199-201, 213), which depends on the material alone.
static int2 material_type_entry(const pt_material &m)
{
    const float inv_coef = 1.f / m.coef;
    int bits;
    memcpy(&bits, &inv_coef, 4);
    return make_int2(m.type, bits);
}

int pt_init_scene(pt_handle h, const pt_material *materials, uint32_t n_materials,
                  const pt_sphere *spheres, uint32_t n_spheres,
                  const pt_light *lights, uint32_t n_lights)
{
    return pt_init_scene_triangles(h, materials, n_materials, spheres, n_spheres, nullptr, 0, lights, n_lights);
}

// Device form of a triangle: (v0, material id), (v1 - v0, light id), (v2 - v0, -), (unit normal, -).  The edges
// are the single subtractions the intersection test starts with, the normal is normalize(e1 x e2) evaluated
// like the reference's Vec::Cross / Normalize (geometry.h:45-47,131-133: a * (1.f / Length())) -- this file is
// compiled without contraction, sqrtf and '/' are IEEE -- so hoisting them to the host changes no bit.
static void triangle_entry(const pt_triangle &t, float4 out[4])
{
    const float e1[3] = {t.v1[0] - t.v0[0], t.v1[1] - t.v0[1], t.v1[2] - t.v0[2]};
    const float e2[3] = {t.v2[0] - t.v0[0], t.v2[1] - t.v0[1], t.v2[2] - t.v0[2]};
    const float c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const float len = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const float inv = 1.f / len;
    float mbits, lbits;
    memcpy(&mbits, &t.material_id, 4);
    memcpy(&lbits, &t.light_id, 4);
    out[0] = make_float4(t.v0[0], t.v0[1], t.v0[2], mbits);
    out[1] = make_float4(e1[0], e1[1], e1[2], lbits);
    out[2] = make_float4(e2[0], e2[1], e2[2], 0.f);
    out[3] = make_float4(c[0] * inv, c[1] * inv, c[2] * inv, 0.f);
}

int pt_init_scene_triangles(pt_handle h, const pt_material *materials, uint32_t n_materials,
                            const pt_sphere *spheres, uint32_t n_spheres,
                            const pt_triangle *triangles, uint32_t n_triangles,
                            const pt_light *lights, uint32_t n_lights)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if ((n_materials && !materials) || (n_spheres && !spheres) || (n_lights && !lights) || (n_triangles && !triangles))
        return fail(h, PT_ERR_INVALID, "pt_init_scene: NULL array with non-zero count");
    if (n_spheres > 0x7fffffffu / 2 || n_triangles > 0x7fffffffu / 2 - n_spheres)
        return fail(h, PT_ERR_INVALID, "pt_init_scene: too many primitives");
    for (uint32_t i = 0; i < n_triangles; ++i) {
        if (triangles[i].material_id >= n_materials)
            return fail(h, PT_ERR_INVALID, "pt_init_scene: triangle %u references material %u of %u", i,
                        triangles[i].material_id, n_materials);
        if (triangles[i].light_id >= (int32_t)n_lights)
            return fail(h, PT_ERR_INVALID, "pt_init_scene: triangle %u references light %d of %u", i,
                        triangles[i].light_id, n_lights);
    }
    for (uint32_t i = 0; i < n_spheres; ++i) {
        if (spheres[i].material_id >= n_materials)
            return fail(h, PT_ERR_INVALID, "pt_init_scene: sphere %u references material %u of %u", i,
                        spheres[i].material_id, n_materials);
        if (spheres[i].light_id >= (int32_t)n_lights)
            return fail(h, PT_ERR_INVALID, "pt_init_scene: sphere %u references light %d of %u", i,
                        spheres[i].light_id, n_lights);
    }
    for (uint32_t i = 0; i < n_lights; ++i)
        if (lights[i].type == PT_AREA_LIGHT && lights[i].u.prim_id >= n_spheres + n_triangles)
            return fail(h, PT_ERR_INVALID, "pt_init_scene: light %u sits on primitive %u of %u", i, lights[i].u.prim_id,
                        n_spheres + n_triangles);
    if (h->async_series) PT_HIP(h, join_async(h));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    free_scene(h);

    // Repack the reference's AoS Primitive (primitive.h:77-81) to SoA float4 pairs.  r*r and
    // 1/r are the same single IEEE operations the reference performs per test / per hit
    // (primitive.h:42,74), so hoisting them to the host does not change a bit.
    std::vector<float4> sph(n_spheres ? n_spheres : 1), aux(n_spheres ? n_spheres : 1);
    for (uint32_t i = 0; i < n_spheres; ++i) {
        const pt_sphere &s = spheres[i];
        sph[i] = make_float4(s.pos[0], s.pos[1], s.pos[2], s.radius * s.radius);
        float mbits, lbits;
        memcpy(&mbits, &s.material_id, 4);
        memcpy(&lbits, &s.light_id, 4);
        aux[i] = make_float4(1.f / s.radius, mbits, lbits, s.radius);
    }
    std::vector<float4> mat(n_materials ? n_materials : 1), light(n_lights ? n_lights : 1);
    std::vector<int2> mtype(n_materials ? n_materials : 1);
    for (uint32_t i = 0; i < n_materials; ++i) {
        mat[i] = material_table_entry(materials[i]);
        mtype[i] = material_type_entry(materials[i]);
    }
    std::vector<float4> light_geom(n_lights ? n_lights : 1);
    for (uint32_t i = 0; i < n_lights; ++i) {
        light[i] = make_float4(lights[i].intensity[0], lights[i].intensity[1], lights[i].intensity[2], 0.f);
        // what SampleLight reads (scene.h:150-170): the position, or the sphere the light sits on
        const int32_t type = lights[i].type;
        float tbits, pbits;
        memcpy(&tbits, &type, 4);
        memcpy(&pbits, &lights[i].u.prim_id, 4);
        // a light of any other type samples to black (scene.h:169) and keeps only its type here:
        // the device never reads a position or a sphere index out of it
        light_geom[i] = type == PT_AREA_LIGHT ? make_float4(pbits, 0.f, 0.f, tbits)
                      : type == PT_POINT_LIGHT ? make_float4(lights[i].u.pos[0], lights[i].u.pos[1], lights[i].u.pos[2], tbits)
                                               : make_float4(0.f, 0.f, 0.f, tbits);
    }

    // a failure half-way leaves no partial scene behind
    auto upload = [](auto **dst, const auto &src) -> hipError_t {
        const size_t bytes = src.size() * sizeof(src[0]);
        hipError_t e = hipMalloc((void **)dst, bytes);
        return e != hipSuccess ? e : hipMemcpy(*dst, src.data(), bytes, hipMemcpyHostToDevice);
    };
    hipError_t ue = upload(&h->d_sph, sph);
    if (ue == hipSuccess) ue = upload(&h->d_aux, aux);
    if (ue == hipSuccess) ue = upload(&h->d_mat, mat);
    if (ue == hipSuccess) ue = upload(&h->d_light, light);
    if (ue == hipSuccess) ue = upload(&h->d_mat_type, mtype);
    if (ue == hipSuccess) ue = upload(&h->d_light_geom, light_geom);
    std::vector<float4> tri((size_t)n_triangles * 4);
    for (uint32_t i = 0; i < n_triangles; ++i) triangle_entry(triangles[i], &tri[(size_t)i * 4]);
    if (ue == hipSuccess && n_triangles) ue = upload(&h->d_tri, tri);
    if (ue != hipSuccess) {
        free_scene(h);
        return fail(h, ue == hipErrorOutOfMemory ? PT_ERR_NOMEM : PT_ERR_DEVICE, "pt_init_scene: scene upload failed: %s",
                    hipGetErrorString(ue));
    }
    h->n_prims = n_spheres; h->n_mats = n_materials; h->n_lights = n_lights;
    h->n_tris = n_triangles;
    h->scene_bytes = (sph.size() + aux.size() + mat.size() + light.size() + light_geom.size() + tri.size()) * sizeof(float4) +
                     mtype.size() * sizeof(int);
    if (n_spheres > kBvhMinSpheres && n_triangles == 0) {   // scenes with triangles are rendered by the linear scan
        const int rc = upload_bvh(h, sph, spheres, n_spheres);
        if (rc != PT_OK) { free_scene(h); return rc; }
    }
    // fingerprint for checkpoints; of a light only the union member its type selects counts
    h->scene_hash = fingerprint(spheres, n_spheres * sizeof(pt_sphere), fingerprint(materials, n_materials * sizeof(pt_material)));
    if (n_triangles) h->scene_hash = fingerprint(triangles, n_triangles * sizeof(pt_triangle), h->scene_hash);
    for (uint32_t i = 0; i < n_lights; ++i) {
        h->scene_hash = fingerprint(&lights[i].type, sizeof lights[i].type, h->scene_hash);
        if (lights[i].type == PT_POINT_LIGHT) h->scene_hash = fingerprint(lights[i].u.pos, sizeof lights[i].u.pos, h->scene_hash);
        else if (lights[i].type == PT_AREA_LIGHT) h->scene_hash = fingerprint(&lights[i].u.prim_id, sizeof(uint32_t), h->scene_hash);
        h->scene_hash = fingerprint(lights[i].intensity, sizeof lights[i].intensity, h->scene_hash);
    }
    h->scene_ready = true;
    return PT_OK;
}

int pt_set_camera(pt_handle h, const pt_camera_pod *camera)
{
    if (!h) return PT_ERR_INVALID;
    if (!camera_ok(camera)) return fail(h, PT_ERR_INVALID, "pt_set_camera: bad camera");
    if (camera->width != h->camera.width || camera->height != h->camera.height)
        return fail(h, PT_ERR_INVALID, "pt_set_camera: image size is latched at pt_create (%ux%u)",
                    h->camera.width, h->camera.height);
    h->camera = *camera;
    return PT_OK;
}

int pt_set_stream(pt_handle h, void *hip_stream)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (h->async_series) PT_HIP(h, join_async(h));   // on the stream the series was started on
    h->stream = (hipStream_t)hip_stream;
    return PT_OK;
}

int pt_set_accum_mode(pt_handle h, int mode)
{
    if (!h) return PT_ERR_INVALID;
    if (mode != PT_ACCUM_RUNNING_MEAN && mode != PT_ACCUM_SUM)
        return fail(h, PT_ERR_INVALID, "pt_set_accum_mode: unknown mode %d", mode);
    h->accum_mode = mode;
    return PT_OK;
}

int pt_set_iteration(pt_handle h, uint32_t iteration)
{
    if (!h) return PT_ERR_INVALID;
    h->iteration = iteration;
    return PT_OK;
}

int pt_set_row_bands(pt_handle h, uint32_t band_rows, uint32_t rank, uint32_t world)
{
    if (!h) return PT_ERR_INVALID;
    if (world == 0 || rank >= world) return fail(h, PT_ERR_INVALID, "pt_set_row_bands: rank %u of %u", rank, world);
    if (world > 1 && (band_rows == 0 || band_rows % 16 != 0))
        return fail(h, PT_ERR_INVALID, "pt_set_row_bands: band_rows %u must be a positive multiple of 16", band_rows);
    h->band_rows = world > 1 ? band_rows : 0;
    h->band_rank = rank;
    h->band_world = world;
    return PT_OK;
}

int pt_set_integrator(pt_handle h, int integrator)
{
    if (!h) return PT_ERR_INVALID;
    if (integrator != PT_INTEGRATOR_PATH && integrator != PT_INTEGRATOR_NEE_MIS)
        return fail(h, PT_ERR_INVALID, "pt_set_integrator: unknown integrator %d", integrator);
    h->integrator = integrator;
    return PT_OK;
}

int pt_set_accel(pt_handle h, int mode)
{
    if (!h) return PT_ERR_INVALID;
    if (mode != PT_ACCEL_AUTO && mode != PT_ACCEL_BRUTE_FORCE)
        return fail(h, PT_ERR_INVALID, "pt_set_accel: unknown mode %d", mode);
    h->accel = mode;
    return PT_OK;
}

uint32_t pt_accel_nodes(pt_handle h) { return h ? h->bvh.n_nodes : 0; }

int pt_debug_build_bvh(const pt_sphere *spheres, uint32_t n_spheres, uint32_t counts[4], float *nodes,
                       int32_t *leaf_index, int32_t *big_index)
{
    if (!counts || (n_spheres && !spheres)) return PT_ERR_INVALID;
    std::vector<float4> sph(n_spheres);
    std::vector<float> radius(n_spheres);
    for (uint32_t i = 0; i < n_spheres; ++i) {
        sph[i] = make_float4(spheres[i].pos[0], spheres[i].pos[1], spheres[i].pos[2], spheres[i].radius * spheres[i].radius);
        radius[i] = spheres[i].radius;
    }
    BvhData d;
    const bool built = build_bvh(sph.data(), radius.data(), n_spheres, &d);
    const bool fill = built && counts[0] == d.nodes.size() / kBvhNodeVec4 && counts[1] == d.leaf_index.size() / kBvhLeafSize &&
                      counts[2] == d.big_index.size();
    if (fill) {
        if (nodes) memcpy(nodes, d.nodes.data(), d.nodes.size() * sizeof(float4));
        if (leaf_index) memcpy(leaf_index, d.leaf_index.data(), d.leaf_index.size() * sizeof(int));
        if (big_index && !d.big_index.empty()) memcpy(big_index, d.big_index.data(), d.big_index.size() * sizeof(int));
    }
    counts[0] = built ? (uint32_t)(d.nodes.size() / kBvhNodeVec4) : 0;
    counts[1] = built ? (uint32_t)(d.leaf_index.size() / kBvhLeafSize) : 0;
    counts[2] = built ? (uint32_t)d.big_index.size() : 0;
    counts[3] = built ? d.depth : 0;
    return PT_OK;
}

int pt_debug_bvh_grid(const pt_sphere *spheres, uint32_t n_spheres, float grid[6], uint32_t *qnodes, uint32_t n_nodes)
{
    if (!grid || !qnodes || (n_spheres && !spheres)) return PT_ERR_INVALID;
    std::vector<float4> sph(n_spheres);
    std::vector<float> radius(n_spheres);
    for (uint32_t i = 0; i < n_spheres; ++i) {
        sph[i] = make_float4(spheres[i].pos[0], spheres[i].pos[1], spheres[i].pos[2], spheres[i].radius * spheres[i].radius);
        radius[i] = spheres[i].radius;
    }
    BvhData d;
    if (!build_bvh(sph.data(), radius.data(), n_spheres, &d)) return PT_ERR_STATE;
    if (d.qnodes.size() != (size_t)n_nodes * 4) return PT_ERR_INVALID;
    for (int a = 0; a < 3; ++a) { grid[a] = d.grid_origin[a]; grid[3 + a] = d.grid_step[a]; }
    memcpy(qnodes, d.qnodes.data(), d.qnodes.size() * sizeof(uint4));
    return PT_OK;
}

int pt_set_accel_budget(pt_handle h, uint32_t visits)
{
    if (!h) return PT_ERR_INVALID;
    if (!h->scene_ready) return fail(h, PT_ERR_STATE, "pt_set_accel_budget before pt_init_scene");
    h->bvh.budget = visits ? visits : h->bvh_default_budget;
    return PT_OK;
}

int pt_accel_stats(pt_handle h, uint64_t out[32])
{
    if (!h || !out) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    memset(out, 0, 32 * sizeof(uint64_t));
    if (!h->d_bvh) return PT_OK;
    if (h->async_series) PT_HIP(h, join_async(h));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    PT_HIP(h, hipMemcpy(out, h->bvh.stats, 32 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    PT_HIP(h, hipMemset(h->bvh.stats, 0, 32 * sizeof(uint64_t)));
    return PT_OK;
}

int pt_lane_stats(pt_handle h, uint64_t out[64])
{
    if (!h || !out) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (h->async_series) PT_HIP(h, join_async(h));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    unsigned long long v[64];
    PT_HIP(h, read_lane_stats(v));
    PT_HIP(h, add_lane_stats_wavefront(v));
    PT_HIP(h, add_lane_stats_nee(v));
    for (int i = 0; i < 64; ++i) out[i] = (uint64_t)v[i];
    return PT_OK;
}

int pt_set_kernel_variant(pt_handle h, int variant)
{
    if (!h) return PT_ERR_INVALID;
    if (variant != PT_VARIANT_AUTO && variant != PT_VARIANT_REGENERATION && variant != PT_VARIANT_CLASSIC &&
        variant != PT_VARIANT_WAVEFRONT && variant != PT_VARIANT_STEALING)
        return fail(h, PT_ERR_INVALID, "pt_set_kernel_variant: unknown variant %d", variant);
    // every tuning knob back to its default
    h->small_scene_opts = 1;
    h->chunks = 0;
    h->steal_unit = 0;
    h->wf_dynamic_fetch = 32;
    h->regen_min_idle = variant == PT_VARIANT_CLASSIC ? 64 : 1;
    h->auto_variant = variant == PT_VARIANT_AUTO;
    h->wavefront = variant == PT_VARIANT_WAVEFRONT;
    h->steal = variant == PT_VARIANT_AUTO || variant == PT_VARIANT_STEALING;
    return PT_OK;
}

int pt_debug_set_tuning(pt_handle h, int key, int value)
{
    if (!h) return PT_ERR_INVALID;
    switch (key) {
    case PT_TUNE_REGEN_MIN_IDLE:
        if (value < 1 || value > 64) break;
        h->regen_min_idle = value;
        return PT_OK;
    case PT_TUNE_CAMERA_RAY_RING:
        if (value != 0 && value != 1) break;
        h->small_scene_opts = value;
        return PT_OK;
    case PT_TUNE_STEAL_UNIT:
        if (value != 0 && value != 1 && value != 4) break;
        h->steal_unit = value;
        return PT_OK;
    case PT_TUNE_ITERATION_RANGES:
        if (value < 0 || value > 32) break;
        h->chunks = value;
        return PT_OK;
    case PT_TUNE_BVH_NODE_STEPS:
        if (value < 1 || value > 16) break;
        h->bvh.node_steps = (uint32_t)value;
        return PT_OK;
    case PT_TUNE_BVH_STACK:
        if (value < 0 || value > 64) break;
        if (h->bvh.n_nodes) h->bvh.depth = value ? (uint32_t)value : h->bvh_default_stack;
        return PT_OK;
    case PT_TUNE_WAVEFRONT_REFILL:
        if (value < 0 || value > 64) break;
        h->wf_dynamic_fetch = value;
        return PT_OK;
    case PT_TUNE_WAVEFRONT_BATCH:
        if (value < 0 || value > 4096) break;
        h->wf_target_mpaths = (uint32_t)value;
        return PT_OK;
    case PT_TUNE_POOL_ITERATIONS:
        if (value < 0) break;
        h->pool_max_iterations = value;
        return PT_OK;
    case PT_TUNE_WAVEFRONT_WHOLE_PATHS:
        if (value < -1 || value > 2) break;
        h->wf_whole_paths = value;
        return PT_OK;
    case PT_TUNE_POOL_WORKGROUPS:
        if (value < 0 || value > 8) break;
        h->pool_workgroups = value;
        return PT_OK;
    case PT_TUNE_BUFFER_SETS:
        if (value != 0 && value != 1) break;
        h->wf_buffer_sets = value;
        return PT_OK;
    default:
        return fail(h, PT_ERR_INVALID, "pt_debug_set_tuning: unknown key %d", key);
    }
    return fail(h, PT_ERR_INVALID, "pt_debug_set_tuning: value %d out of range for key %d", value, key);
}

static uint32_t div_up(uint32_t a, uint32_t b);
static uint32_t auto_chunks(uint64_t tiles, uint32_t n_iterations);
static uint32_t wavefront_batch_iterations(const pt_context *h, int mode, uint64_t paths_per_it, uint32_t n_iterations, uint32_t *sets);
static int wavefront_mode(const pt_context *h, uint32_t n_iterations);
static bool takes_wavefront(const pt_context *h, uint64_t last_iteration, uint32_t n_iterations);
static int wavefront_whole_paths(const pt_context *h);
// Rows of a frame of `height` rows an instance renders: all of them, or those of its row bands
// (bands of band_rows rows, band b to rank b % world).
static uint32_t owned_rows(uint32_t height, uint32_t band_rows, uint32_t rank, uint32_t world)
{
    if (world <= 1) return height;
    uint32_t rows = 0;
    for (uint64_t y = (uint64_t)rank * band_rows; y < height; y += (uint64_t)band_rows * world)
        rows += height - y < band_rows ? (uint32_t)(height - y) : band_rows;
    return rows;
}
static uint32_t owned_rows(const pt_context *h) { return owned_rows(h->camera.height, h->band_rows, h->band_rank, h->band_world); }

// Slots per sub-queue for a batch of `iterations` iterations over a w x h rectangle: the batch is
// dealt out in blocks of 256 paths, block k to sub-queue k % 64, and a sub-queue keeps its survivors.
static uint64_t wavefront_sub_capacity(uint32_t w, uint32_t h, uint64_t iterations)
{
    const uint64_t blocks = ((uint64_t)w * h * 4 * iterations + 255) / 256;
    return (blocks + kWfSubQueues - 1) / kWfSubQueues * 256;
}
static uint64_t wavefront_bytes_per_path(int mode);
static uint64_t wavefront_mask_bytes(uint64_t paths);
static uint64_t wavefront_queues(int mode);

int pt_set_memory_limit(pt_handle h, uint64_t bytes)
{
    if (!h) return PT_ERR_INVALID;
    h->memory_limit = bytes;
    h->steal_alloc_failed = false;   // a raised limit may allow the scratch pool after all
    return PT_OK;
}

int pt_memory_footprint(pt_handle h, uint32_t n_iterations, uint64_t bytes[2])
{
    if (!h || !bytes) return PT_ERR_INVALID;
    const uint64_t frame = (uint64_t)h->camera.width * h->camera.height * 3 * sizeof(float);
    const uint64_t fixed = sizeof(unsigned long long) + h->scene_bytes + (h->d_internal_buffer ? frame : 0);
    const uint64_t steal_now = h->d_steal ? kStealPoolBytes + kStealMaskBytes : 0;
    const uint64_t tickets_now = (uint64_t)h->tickets_capacity * sizeof(uint32_t);
    const uint64_t counts = (uint64_t)(h->max_bounces + 2) * kWfSubQueues * kWfCountStride * sizeof(uint32_t);
    const uint64_t queue_bytes = 2 * sizeof(float4) + sizeof(WavefrontTail);
    const uint64_t wf_set = (uint64_t)h->wf_n_queues * h->wf_sub_capacity * kWfSubQueues * queue_bytes +
                            h->wf_capacity * 3 * sizeof(float) + counts + h->wf_mask_bytes;
    const uint64_t wf_now = ((h->wf_pipe[0].counts ? 1 : 0) + (h->wf_pipe[1].counts ? 1 : 0)) * wf_set;
    bytes[0] = fixed + steal_now + tickets_now + wf_now;
    // what the next render adds: wavefront queues of one batch, or the scratch pool and tickets
    uint64_t steal_next = steal_now, tickets_next = tickets_now, wf_next = wf_now;
    const uint64_t paths_per_it = (uint64_t)h->camera.width * owned_rows(h) * 4;
    uint32_t wf_sets = 1;
    const int wf_mode = wavefront_mode(h, n_iterations);
    const uint32_t batch_it = n_iterations && paths_per_it ? wavefront_batch_iterations(h, wf_mode, paths_per_it, n_iterations, &wf_sets) : 0;
    if (n_iterations && h->scene_ready && paths_per_it && takes_wavefront(h, h->iteration + (uint64_t)n_iterations, n_iterations) && batch_it > 0) {
        const uint64_t capacity = paths_per_it * batch_it;
        const uint64_t sub_cap = wavefront_sub_capacity(h->camera.width, owned_rows(h), batch_it);
        const uint64_t n_queues = wavefront_queues(wf_mode);
        const uint64_t need = wf_sets * (n_queues * sub_cap * kWfSubQueues * queue_bytes + capacity * 3 * sizeof(float) + counts +
                                         wavefront_mask_bytes(capacity));
        if (need > wf_next) wf_next = need;
    } else if (n_iterations && h->scene_ready && !h->wavefront && h->steal && !h->steal_alloc_failed &&
               (!h->memory_limit || kStealPoolBytes + kStealMaskBytes <= h->memory_limit)) {
        steal_next = kStealPoolBytes + kStealMaskBytes;
        const uint64_t tiles = (uint64_t)div_up(h->camera.width, 16) * div_up(h->camera.height, 16);
        uint32_t n_chunks = h->chunks > 0 ? (uint32_t)h->chunks : auto_chunks(tiles, n_iterations);
        if (n_chunks > n_iterations / 4) n_chunks = n_iterations / 4;
        if (n_chunks > 1 && (1 + tiles) * sizeof(uint32_t) > tickets_next) tickets_next = (1 + tiles) * sizeof(uint32_t);
    }
    bytes[1] = fixed + steal_next + tickets_next + wf_next;
    return PT_OK;
}

int pt_reset(pt_handle h)
{
    if (!h) return PT_ERR_INVALID;
    h->iteration = 0;
    h->fresh = true;
    return PT_OK;
}

uint32_t pt_iteration(pt_handle h) { return h ? h->iteration : 0; }

static void fill_camera(const pt_context *h, CameraParams &c)
{
    for (int i = 0; i < 3; ++i) {
        c.pos[i] = h->camera.pos[i];
        c.pxx[i] = h->camera.px_x[i];
        c.pxy[i] = h->camera.px_y[i];
        c.first[i] = h->camera.first_ray_dir[i];
    }
    c.width = h->camera.width; c.height = h->camera.height;
}

static void fill_scene(const pt_context *h, SceneParams &s)
{
    s.sph = h->d_sph; s.aux = h->d_aux; s.mat = h->d_mat; s.mat_type = h->d_mat_type; s.light = h->d_light;
    s.light_geom = h->d_light_geom;
    s.n_prims = h->n_prims; s.n_mats = h->n_mats; s.n_lights = h->n_lights;
    s.tri = h->d_tri; s.n_tris = h->n_tris;
    if (h->accel == PT_ACCEL_AUTO) s.bvh = h->bvh;
    else memset(&s.bvh, 0, sizeof s.bvh);
}

static void fill_params(const pt_context *h, RenderParams &p)
{
    memset(&p, 0, sizeof p);
    fill_camera(h, p.cam);
    fill_scene(h, p.scene);
    p.seed = h->seed;
    p.max_bounces = h->max_bounces;
    p.regen_min_idle = h->regen_min_idle;
    p.small_scene_opts = h->small_scene_opts;
}

static uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

// Device memory per path in flight of the wavefront variant: the path's radiance + two queues of
// 44-byte path records (one launch per bounce), one queue (camera rays in a launch of their own,
// then whole paths), or none (whole paths from the camera ray on).
static const uint64_t kWfQueueRecord = 2 * sizeof(float4) + sizeof(WavefrontTail);
static const uint64_t kWfBytesPerPath = 2 * kWfQueueRecord + 3 * sizeof(float);
// Device memory the wavefront buffers are sized for by default (two sets together).
// Measured on the 10 007-sphere scene, 1080p x 16 iterations, per-bounce launches: 16 Mi paths in
// flight 111 ms, 32 Mi 97, 64 Mi 92.7, 128 Mi 90.8 -- every launch ends with a tail of its longest
// traversals, and the fewer and larger the launches the less that costs.
static const uint64_t kWfTargetBytes = (64ull << 20) * kWfBytesPerPath;

// BVH scenes traced by the persistent waves: how the bounces map to launches.
//   0  one launch per bounce over the queues (100 bytes per path in flight)
//   1  one launch per batch follows every path from its camera ray to its end (no queues, 12 bytes)
//   2  the camera rays in a launch of their own, a second launch follows every surviving path from
//      its first bounce to its end (one queue, 56 bytes)
// Measured on the 10 007-sphere scene, 1080p x 16 iterations (ms, HBM bytes per render):
// mode 0 91.5 ms, 38.8 GB; mode 2 92.2 ms, 12.7 GB; mode 1 96.3 ms, 2.3 GB.
// Camera rays are coherent and trace at twice the rate of scattered rays as long as they share
// waves with each other, which mode 1 gives up; a queue also keeps the scattered rays in the runs
// a wave appended together, which is worth the last 0.7 % of mode 0.  The default is mode 2: a
// third of the traffic and 56 instead of 100 bytes per path in flight for that 0.7 %; mode 1 when
// the memory limit has no room for one iteration's queue.
static int wavefront_whole_paths(const pt_context *h)
{
    if (h->wf_dynamic_fetch <= 0 || h->accel != PT_ACCEL_AUTO || h->bvh.n_nodes == 0) return 0;
    // the direct-light integrator carries more per path than the queue records hold (the parked term,
    // the last BSDF pdf): its paths stay in their lanes from the camera ray on
    if (h->integrator == PT_INTEGRATOR_NEE_MIS) return 1;
    if (h->wf_whole_paths >= 0) return h->wf_whole_paths;
    const uint64_t paths_per_it = (uint64_t)h->camera.width * h->camera.height * 4;
    if (h->memory_limit && h->memory_limit / (kWfQueueRecord + 3 * sizeof(float) + 1) < paths_per_it) return 1;
    return 2;
}

// The pool and wavefront kernels turn path numbers into rows, bands and iterations with a division
// through the float reciprocal (div_small in pt_wavefront.hip), exact while every quotient stays
// below 2^20 and every divisor below 2^30: rows of the frame, iterations of a batch (capped in
// plan_batches), band_rows * world.  Frames outside that go through the megakernel.
static bool wavefront_indexing_ok(uint32_t height, uint32_t band_rows, uint32_t band_world)
{
    if (height >= (1u << 20)) return false;
    return band_world <= 1 || (uint64_t)band_rows * band_world < (1ull << 30);
}
static bool wavefront_indexing_ok(const pt_context *h)
{
    return wavefront_indexing_ok(h->camera.height, h->band_rows, h->band_world);
}

// Launches over a scene that is rendered by the linear scan (resident in LDS or streamed through
// it in tiles): the path pool (wf_pool_kernel) instead of the megakernel, whose lanes own pixels.
static bool takes_pool(const pt_context *h, uint64_t last_iteration, uint32_t n_iterations)
{
    SceneParams s;
    fill_scene(h, s);
    return h->auto_variant && s.bvh.n_nodes == 0 && (h->n_tris == 0 || h->integrator == PT_INTEGRATOR_PATH) &&
           n_iterations > 0 && n_iterations <= (uint32_t)h->pool_max_iterations &&
           last_iteration < (1u << 30) && wavefront_indexing_ok(h);
}

// How a render of n_iterations maps to launches: 0..2 as wavefront_whole_paths, 3: the path pool.
static int wavefront_mode(const pt_context *h, uint32_t n_iterations)
{
    if (!h->wavefront && takes_pool(h, h->iteration + (uint64_t)n_iterations, n_iterations)) return 3;
    return wavefront_whole_paths(h);
}

// Beside queues and radiance, one mask byte per pixel-iteration (which of its four paths have an
// entry in the radiance buffer): a quarter of a byte per path, planned as one.
static uint64_t wavefront_bytes_per_path(int mode)
{
    return (mode == 0 ? 2 : mode == 2 ? 1 : 0) * kWfQueueRecord + 3 * sizeof(float) + 1;
}
static uint64_t wavefront_mask_bytes(uint64_t paths) { return (paths / 4 + 3) / 4 * 4; }
static uint64_t wavefront_queues(int mode) { return mode == 0 ? 2 : mode == 2 ? 1 : 0; }

// Iterations per wavefront batch of a frame with `paths_per_it` paths per iteration, within the
// handle's memory limit; 0 when not even one iteration fits.  A render of more iterations than
// that runs as several batches alternating between two buffer sets of half the size each
// (*sets = 2); when half the memory does not hold one iteration, one set takes all of it.
// Paths in flight the batch buffers (both sets together) are planned for.
static uint64_t plan_target(uint64_t per_path, uint64_t memory_limit, uint32_t target_mpaths)
{
    uint64_t target = target_mpaths ? (uint64_t)target_mpaths << 20 : kWfTargetBytes / per_path;
    if (memory_limit && memory_limit / per_path < target) target = memory_limit / per_path;
    return target;
}

static uint32_t plan_batches(uint64_t paths_per_it, uint32_t n_iterations, uint64_t per_path, uint64_t memory_limit,
                             uint32_t target_mpaths, uint32_t *sets, bool one_set = false)
{
    const uint64_t target = plan_target(per_path, memory_limit, target_mpaths);
    uint32_t n_sets = 2;
    uint64_t batch_it = target / 2 / paths_per_it;
    if (batch_it < 1 || batch_it >= n_iterations || one_set) {   // one_set: the same batches, one after the other
        n_sets = 1;
        batch_it = batch_it < 1 ? target / paths_per_it : batch_it;
        if (batch_it < 1) batch_it = memory_limit ? 0 : 1;
    }
    if (batch_it > n_iterations) batch_it = n_iterations;
    // the kernels number the paths of a batch with 32 bits, and find a path's iteration within the
    // batch by a division that is exact for quotients below 2^20
    if (batch_it > 0xffffffffull / paths_per_it) batch_it = 0xffffffffull / paths_per_it;
    if (batch_it > (1u << 20) - 1) batch_it = (1u << 20) - 1;
    if (sets) *sets = n_sets;
    return (uint32_t)batch_it;
}

static uint32_t wavefront_target_mpaths(const pt_context *h, int mode)
{
    return h->wf_target_mpaths ? h->wf_target_mpaths : mode == 3 ? 266u : 0;
}

// Path-pool launches of a pt_run_async series: persistent workgroups per CU.  A launch of few
// paths is mostly start-up and tail; with about half the resident set per launch, two consecutive
// calls' launches are resident side by side and one's tail runs under the other's body.  Measured
// at 1080p, demo scene, 64 calls of one iteration + one wait, workgroups per CU 2..8: 29.3, 25.1,
// 25.3, 25.0, 28.2, 28.2, 28.2 ms (one launch of 64 iterations: 24.5 ms); 32 calls of two: 28.8,
// 24.7, 25.0, 24.8, 26.2, 26.2, 26.2; 16 calls of four: 5: 24.6, 8: 25.1; the 71-sphere scene at
// depth 16: 64 calls 5: 62.9, 8: 63.4, 16 calls of four 5: 62.1, 8: 61.4.  So: five for calls of
// up to 20 Mi paths (two iterations at 1080p), the full set beyond.
static int async_pool_workgroups(uint64_t paths_per_call)
{
    return paths_per_call <= (20ull << 20) ? 5 : 0;
}

// pt_run_async: can a call of n_iterations be ONE batch in one of two buffer sets, so that
// consecutive calls alternate between the sets?  (Half the planned paths in flight hold it.)
static bool async_call_alternates(const pt_context *h, int mode, uint64_t paths_per_it, uint32_t n_iterations)
{
    if (h->wf_buffer_sets == 1 || n_iterations >= (1u << 20)) return false;
    const uint64_t half = plan_target(wavefront_bytes_per_path(mode), h->memory_limit, wavefront_target_mpaths(h, mode)) / 2;
    return half / paths_per_it >= n_iterations && (uint64_t)n_iterations * paths_per_it <= 0xffffffffull;
}

static uint32_t wavefront_batch_iterations(const pt_context *h, int mode, uint64_t paths_per_it, uint32_t n_iterations, uint32_t *sets = nullptr)
{
    // The path pool: 256 Mi paths in flight over the two sets (3.3 GB; 16 iterations per batch at
    // 1080p).  Measured on the 1080p x 64 demo render, iterations per batch: 32 24.57 ms, 16 24.57,
    // 8 24.69, 4 25.17, 2 26.34 -- the resolve reads only the entries of the paths that met a light,
    // so a batch costs little beyond its launches.
    const uint32_t target_mpaths = wavefront_target_mpaths(h, mode);
    return plan_batches(paths_per_it, n_iterations, wavefront_bytes_per_path(mode), h->memory_limit, target_mpaths, sets,
                        h->wf_buffer_sets == 1);
}

int pt_debug_plan_batches(uint32_t width, uint32_t height, uint32_t band_rows, uint32_t band_rank, uint32_t band_world,
                          uint32_t n_iterations, uint32_t bytes_per_path, uint64_t memory_limit, uint32_t out[4])
{
    if (!out || !width || !height || !bytes_per_path || (band_world > 1 && (!band_rows || band_rank >= band_world)))
        return PT_ERR_INVALID;
    const uint32_t rows = owned_rows(height, band_rows, band_rank, band_world);
    const uint64_t paths_per_it = (uint64_t)width * rows * 4;
    uint32_t sets = 1;
    const uint32_t batch_it = paths_per_it && n_iterations && wavefront_indexing_ok(height, band_rows, band_world)
                                  ? plan_batches(paths_per_it, n_iterations, bytes_per_path, memory_limit, 0, &sets) : 0;
    out[0] = rows;
    out[1] = batch_it;
    out[2] = batch_it ? sets : 0;
    const uint64_t sub_cap = batch_it ? wavefront_sub_capacity(width, rows, batch_it) : 0;
    out[3] = sub_cap > 0xffffffffull ? 0xffffffffu : (uint32_t)sub_cap;
    return PT_OK;
}

// `sets` sets of queues, per-path radiance and counters for batches of `capacity` paths (kept while
// they suffice).  A failed allocation frees what was allocated and reports PT_ERR_NOMEM.
static int ensure_wavefront(pt_context *h, int mode, uint64_t capacity, uint64_t sub_cap64, uint32_t sets, size_t *counts_bytes_out)
{
    const int n_queues = (int)wavefront_queues(mode);
    const uint64_t mask_bytes = wavefront_mask_bytes(capacity);
    if (sub_cap64 > 0xffffff00ull) return fail(h, PT_ERR_INVALID, "wavefront variant: image too large");
    const uint32_t sub_cap = (uint32_t)sub_cap64;
    // the counters, and behind them the radiance mask: both zero between batches
    const size_t counters_bytes = (size_t)(h->max_bounces + 2) * kWfSubQueues * kWfCountStride * sizeof(uint32_t);
    bool have = capacity <= h->wf_capacity && sub_cap <= h->wf_sub_capacity && h->wf_n_queues >= n_queues &&
                mask_bytes <= h->wf_mask_bytes;
    for (uint32_t k = 0; k < sets; ++k) have = have && h->wf_pipe[k].counts;
    const size_t counts_bytes = counters_bytes + (size_t)(have ? h->wf_mask_bytes : mask_bytes);
    *counts_bytes_out = counts_bytes;
    if (sets > 1 && !h->wf_stream2) {
        if (hipStreamCreateWithFlags(&h->wf_stream2, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->wf_fork, hipEventDisableTiming) != hipSuccess)
            return fail(h, PT_ERR_DEVICE, "wavefront variant: cannot create the second stream");
    }
    if (!have) {
        PT_HIP(h, hipStreamSynchronize(h->stream));
        free_wavefront(h);
        const size_t slots = (size_t)sub_cap * kWfSubQueues;
        bool ok = true;
        for (uint32_t k = 0; k < sets && ok; ++k) {
            pt_context::WfPipe &pipe = h->wf_pipe[k];
            for (int i = 0; i < n_queues && ok; ++i) {
                ok = hipMalloc((void **)&pipe.queue[i].a, slots * sizeof(float4)) == hipSuccess &&
                     hipMalloc((void **)&pipe.queue[i].b, slots * sizeof(float4)) == hipSuccess &&
                     hipMalloc((void **)&pipe.queue[i].c, slots * sizeof(WavefrontTail)) == hipSuccess;
            }
            ok = ok && hipMalloc((void **)&pipe.radiance, capacity * 3 * sizeof(float)) == hipSuccess &&
                 hipMalloc((void **)&pipe.counts, counts_bytes) == hipSuccess &&
                 hipEventCreateWithFlags(&pipe.resolved, hipEventDisableTiming) == hipSuccess;
            pipe.counts_dirty = true;
        }
        if (!ok) {
            (void)hipGetLastError();
            free_wavefront(h);
            return fail(h, PT_ERR_NOMEM, "wavefront variant: cannot allocate queues for %u x %llu paths (%llu MiB)", sets,
                        (unsigned long long)capacity, (unsigned long long)(sets * capacity * wavefront_bytes_per_path(mode) >> 20));
        }
        h->wf_capacity = capacity;
        h->wf_sub_capacity = sub_cap;
        h->wf_n_queues = n_queues;
        h->wf_mask_bytes = mask_bytes;
    }
    return PT_OK;
}

static void fill_wavefront(const pt_context *h, WavefrontParams &p)
{
    memset(&p, 0, sizeof p);
    fill_camera(h, p.cam);
    fill_scene(h, p.scene);
    p.seed = h->seed;
    p.max_bounces = h->max_bounces;
    p.dynamic_fetch = h->wf_dynamic_fetch;
    p.sub_capacity = h->wf_sub_capacity;
    p.x0 = 0; p.y0 = 0; p.rect_w = p.cam.width; p.rect_h = p.cam.height;
    p.pool_workgroups_per_cu = h->pool_workgroups;
    p.nee = h->integrator == PT_INTEGRATOR_NEE_MIS;
}

// One batch in buffer set `set` on `stream`: a launch per bounce over the compacted queues, then
// the resolve -- after the resolve of the batch before (`after`, the other set's event, or null):
// the frame buffer takes the batches' iterations in order.
static int launch_wavefront_batch(pt_context *h, WavefrontParams &p, int mode, size_t counts_bytes, int set, hipStream_t stream,
                                  hipEvent_t after, bool record = true, bool zero_radiance = false)
{
    const bool tiled = scene_streams(p.scene);
    pt_context::WfPipe &pipe = h->wf_pipe[set];
    p.counts = pipe.counts;
    p.path_radiance = pipe.radiance;
    p.path_mask = pipe.counts + (size_t)(h->max_bounces + 2) * kWfSubQueues * kWfCountStride;
    // the resolve kernel leaves the counters at zero for the set's next batch
    if (pipe.counts_dirty) PT_HIP(h, hipMemsetAsync(pipe.counts, 0, counts_bytes, stream));
    pipe.counts_dirty = true;
    if (zero_radiance) {   // parity hook of the queue-fed kernels: the radiance buffer is read for every path
        const uint64_t batch_paths = (uint64_t)p.n_iterations * p.rect_w * p.rect_h * 4;
        PT_HIP(h, hipMemsetAsync(pipe.radiance, 0, batch_paths * 3 * sizeof(float), stream));
    }
    const int last = mode == 1 || mode == 3 ? 0 : mode == 2 ? (h->max_bounces < 1 ? h->max_bounces : 1) : h->max_bounces;
#ifdef PT_SORT_EXPERIMENT
    // measurement build: the survivors of every bounce are sorted inside their sub-queues before the next launch
    const char *sort_env = getenv("PT_EXPERIMENT_SORT");
    const int sort_mode = sort_env ? atoi(sort_env) : 0;
    int flip = 0;
#endif
    for (int b = 0; b <= last; ++b) {
        p.bounce = b;
        p.whole_paths = mode == 1 || (mode == 2 && b == 1);
        p.path_pool = mode == 3;
        p.out = pipe.queue[b & 1];          // not written by a whole-path launch
        p.in = pipe.queue[(b & 1) ^ 1];
#ifdef PT_SORT_EXPERIMENT
        if (flip) { p.out = pipe.queue[(b & 1) ^ 1]; p.in = pipe.queue[b & 1]; }
#endif
        PT_HIP(h, launch_wavefront_step(p, tiled, stream));
#ifdef PT_SORT_EXPERIMENT
        if (sort_mode > 0 && b < last && (mode == 0 || mode == 2) && p.scene.bvh.n_nodes) {
            const uint32_t *next_counts = pipe.counts + (size_t)(b + 1) * kWfSubQueues * kWfCountStride;
            PT_HIP(h, sort_queue_experiment(p, p.out, p.in, next_counts, sort_mode, stream));
            flip ^= 1;
        }
#endif
    }
    if (after) PT_HIP(h, hipStreamWaitEvent(stream, after, 0));
    PT_HIP(h, launch_wavefront_resolve(p, stream));
    pipe.counts_dirty = false;
    if (record) PT_HIP(h, hipEventRecord(pipe.resolved, stream));   // for the other set's next resolve
    return PT_OK;
}

// Wavefront variant: batches of whole iterations; per batch one launch per bounce + a resolve.
// Returns PT_ERR_NOMEM (nothing rendered, nothing held) when the queues do not fit.
// async_call (pt_run_async): a call that fits one buffer set is one batch, and consecutive calls
// alternate between the two sets and streams as the batches of one call do; the caller's stream is
// not made to wait for the second stream (join_async does that later).
static int enqueue_wavefront(pt_context *h, float *dbuf, uint32_t n_iterations, bool async_call)
{
    const int mode = wavefront_mode(h, n_iterations);
    const uint32_t rows = owned_rows(h);
    const uint64_t paths_per_it = (uint64_t)h->camera.width * rows * 4;
    if (rows == 0) {   // this rank owns no band of the frame
        h->iteration += n_iterations;
        h->fresh = false;
        return PT_OK;
    }
    if (h->iteration + (uint64_t)n_iterations >= (1u << 30))
        return fail(h, PT_ERR_INVALID, "wavefront variant: iteration numbers must stay below 2^30");
    uint32_t sets = 1;
    uint32_t batch_it = wavefront_batch_iterations(h, mode, paths_per_it, n_iterations, &sets);
    if (batch_it == 0)
        return fail(h, PT_ERR_NOMEM, "wavefront variant: one iteration needs %llu MiB of queues, the memory limit is %llu MiB",
                    (unsigned long long)(paths_per_it * wavefront_bytes_per_path(mode) >> 20), (unsigned long long)(h->memory_limit >> 20));
    const bool alternate = async_call && async_call_alternates(h, mode, paths_per_it, n_iterations);
    if (alternate) { batch_it = n_iterations; sets = 2; }
    else if (h->async_series) PT_HIP(h, join_async(h));   // this call runs in the caller's stream's order
    size_t counts_bytes = 0;
    const int rc = ensure_wavefront(h, mode, paths_per_it * batch_it, wavefront_sub_capacity(h->camera.width, rows, batch_it),
                                    sets, &counts_bytes);
    if (rc != PT_OK) return rc;

    WavefrontParams p;
    fill_wavefront(h, p);
    p.accum_mode = h->accum_mode;
    p.buffer = dbuf;
    p.ray_counter = h->d_ray_counter;
    p.rect_h = rows;
    if (h->band_world > 1) { p.band_rows = h->band_rows; p.band_rank = h->band_rank; p.band_world = h->band_world; }

    if (alternate) {
        // One batch in the set whose turn it is.  The series' first call marks where the caller's stream
        // stood (the second stream starts behind that, once); a call's resolve waits for the resolve of
        // the call before it, which ran in the other set: the frame buffer takes the iterations in order.
        if (!h->async_series) {
            PT_HIP(h, hipEventRecord(h->ev_start, h->stream));
            PT_HIP(h, hipEventRecord(h->wf_fork, h->stream));
            h->async_series = true;
            h->async_forked = false;
            h->async_set = 0;
        }
        const int set = h->async_set;
        if (set == 1 && !h->async_forked) {
            PT_HIP(h, hipStreamWaitEvent(h->wf_stream2, h->wf_fork, 0));
            h->async_forked = true;
        }
        p.first_iteration = h->iteration + 1;
        p.n_iterations = n_iterations;
        p.overwrite = h->accum_mode == PT_ACCUM_RUNNING_MEAN ? (p.first_iteration == 1) : (h->fresh ? 1 : 0);
        p.pool_workgroups_per_cu = h->pool_workgroups ? h->pool_workgroups : async_pool_workgroups(paths_per_it * n_iterations);
        // the other set's event is that of the call before this one, if the series has had one there
        hipEvent_t before = nullptr;
        if (set == 1) before = h->wf_pipe[0].resolved;
        else if (h->async_forked) before = h->wf_pipe[1].resolved;
        const int brc = launch_wavefront_batch(h, p, mode, counts_bytes, set, set ? h->wf_stream2 : h->stream, before, true);
        if (brc != PT_OK) {
            (void)hipStreamSynchronize(h->wf_stream2);
            h->async_series = h->async_on_stream2 = h->async_forked = false;
            h->async_set = 0;
            return brc;
        }
        h->async_on_stream2 = set == 1;
        h->async_set = set ^ 1;
    } else {
        PT_HIP(h, hipEventRecord(h->ev_start, h->stream));
        if (sets > 1) {   // the second stream starts after what the caller's stream holds so far
            PT_HIP(h, hipEventRecord(h->wf_fork, h->stream));
            PT_HIP(h, hipStreamWaitEvent(h->wf_stream2, h->wf_fork, 0));
        }
        uint32_t done = 0;
        int set = 0;
        hipEvent_t before = nullptr;
        while (done < n_iterations) {
            const uint32_t n = n_iterations - done < batch_it ? n_iterations - done : batch_it;
            p.first_iteration = h->iteration + done + 1;
            p.n_iterations = n;
            p.overwrite = h->accum_mode == PT_ACCUM_RUNNING_MEAN ? (p.first_iteration == 1)
                                                                 : ((h->fresh && done == 0) ? 1 : 0);
            const int brc = launch_wavefront_batch(h, p, mode, counts_bytes, set, set ? h->wf_stream2 : h->stream, before, sets > 1);
            if (brc != PT_OK) {   // nothing of this call may still run on the second stream when the caller sees the error
                if (sets > 1) (void)hipStreamSynchronize(h->wf_stream2);
                return brc;
            }
            before = sets > 1 ? h->wf_pipe[set].resolved : nullptr;
            done += n;
            if (sets > 1) set ^= 1;
        }
        // the caller's stream ends behind the last resolve (its own, which waited for the other's, or the other's)
        if (before && before == h->wf_pipe[1].resolved) PT_HIP(h, hipStreamWaitEvent(h->stream, before, 0));
        PT_HIP(h, hipEventRecord(h->ev_stop, h->stream));
    }
    h->timed = true;
    h->last_launch = mode == 3 ? PT_LAUNCH_PATH_POOL : mode == 0 ? PT_LAUNCH_WAVEFRONT : PT_LAUNCH_WAVEFRONT_WHOLE_PATHS;
    h->iteration += n_iterations;
    h->fresh = false;
    return PT_OK;
}

// Number of consecutive iteration ranges a launch of `tiles` workgroup tiles x `n_iterations`
// iterations is cut into (render_steal_kernel).
// Measured on MI355X (demo scene, depth 8, one range -> best): 960x540x64 45 -> 54 G bounces/s
// with 3..6 ranges, 1080p x 64 60 -> 62 with 2..3, 1080p x 256 62.6 -> 66 with 4, no gain at 4K
// (32 400 tiles) and none from ranges shorter than 16 iterations.  So: about 16 rounds of
// workgroups over the chip's ~1536 resident workgroup slots, ranges of at least 16 iterations.
static uint32_t auto_chunks(uint64_t tiles, uint32_t n_iterations)
{
    if (tiles == 0) return 1;
    uint64_t n = (24576 + tiles / 2) / tiles;
    const uint32_t longest = n_iterations / 16 < 8 ? n_iterations / 16 : 8;
    if (n > longest) n = longest;
    return n < 1 ? 1u : (uint32_t)n;
}

// Does this render go through the per-bounce wavefront launches?  Explicitly selected, or
// PT_VARIANT_AUTO on a BVH scene: BVH traversals differ so much in length that the queue-fed
// persistent waves are twice as fast as the megakernel (measured); everything else stays with
// the megakernel.
static bool takes_wavefront(const pt_context *h, uint64_t last_iteration, uint32_t n_iterations)
{
    if (h->wavefront) return true;
    if (takes_pool(h, last_iteration, n_iterations)) return true;
    // (the direct-light integrator too: whole paths from the camera ray on, see wavefront_whole_paths)
    return h->auto_variant && h->accel == PT_ACCEL_AUTO && h->bvh.n_nodes > 0 &&
           (h->integrator == PT_INTEGRATOR_PATH || h->wf_dynamic_fetch > 0) &&
           last_iteration < (1u << 30) && wavefront_indexing_ok(h);
}

// The handle's own scratch of the work-stealing megakernel (pool + occupancy bits in one
// allocation), allocated on first use; false when it cannot be had (the launch then does not steal).
static bool own_steal_scratch(pt_context *h, StealScratch *out)
{
    if (!h->d_steal && !h->steal_alloc_failed) {
        const size_t bytes = kStealPoolBytes + kStealMaskBytes;
        if ((h->memory_limit && bytes > h->memory_limit) || hipMalloc(&h->d_steal, bytes) != hipSuccess) {
            (void)hipGetLastError();
            h->d_steal = nullptr;
            h->steal_alloc_failed = true;
        } else if (hipMemsetAsync((unsigned char *)h->d_steal + kStealPoolBytes, 0, kStealMaskBytes, h->stream) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(h->d_steal);
            h->d_steal = nullptr;
            h->steal_alloc_failed = true;
        }
    }
    if (!h->d_steal) return false;
    out->pool = (float *)h->d_steal;
    out->mask = (unsigned long long *)((unsigned char *)h->d_steal + kStealPoolBytes);
    out->tickets = h->d_tickets;
    out->ticket_words = (uint32_t)(h->tickets_capacity > 0xffffffffu ? 0xffffffffu : h->tickets_capacity);
    return true;
}

// Work stealing and iteration ranges of a megakernel launch of grid_x x grid_y tiles.
// fixed == NULL: the handle's own scratch, allocated (and its ticket array grown) as needed;
// else the statically allocated scratch: nothing is allocated, its occupancy bits and tickets are
// cleared on the stream first.
static int setup_steal(pt_context *h, RenderParams &p, uint32_t grid_x, uint32_t grid_y, uint32_t n_iterations,
                       const StealScratch *fixed)
{
    if (!h->steal || n_iterations >= (1u << 30)) return PT_OK;
    StealScratch sc;
    if (fixed) sc = *fixed;
    else if (!own_steal_scratch(h, &sc)) return PT_OK;
    p.steal_scratch = sc.pool;
    p.steal_mask = sc.mask;
    // chunked launch: more, shorter workgroups (see render_steal_kernel)
    uint32_t n_chunks = h->chunks > 0 ? (uint32_t)h->chunks : auto_chunks((uint64_t)grid_x * grid_y, n_iterations);
    if (n_chunks > n_iterations / 4) n_chunks = n_iterations / 4;   // ranges of at least 4 iterations
    if (n_chunks > 1) n_chunks = div_up(n_iterations, div_up(n_iterations, n_chunks));   // no empty range
    if (n_chunks > 1 && (uint64_t)grid_y * n_chunks > 65535u) n_chunks = 1;
    const size_t words = 1 + (size_t)grid_x * grid_y;
    if (!fixed && n_chunks > 1 && words > h->tickets_capacity) {
        PT_HIP(h, hipStreamSynchronize(h->stream));
        if (h->d_tickets) (void)hipFree(h->d_tickets);
        h->d_tickets = nullptr; h->tickets_capacity = 0;
        // without the tickets the launch is simply not cut into ranges
        if (hipMalloc((void **)&h->d_tickets, words * sizeof(uint32_t)) == hipSuccess)
            h->tickets_capacity = words;
        else
            (void)hipGetLastError();
        sc.tickets = h->d_tickets;
        sc.ticket_words = (uint32_t)h->tickets_capacity;
    }
    // The static array is shared by every handle and stream of the process on this device.  Its
    // occupancy bits are zero when the library is loaded and every workgroup returns the block it took,
    // so they are never cleared here: clearing them under another stream's launch in flight would hand
    // its blocks out a second time.  The tickets hold the last call's counts and must start at zero.
    if (fixed && n_chunks > 1)
        PT_HIP(h, hipMemsetAsync(sc.tickets, 0, words * sizeof(uint32_t), h->stream));
    if (n_chunks > 1 && sc.tickets && words <= sc.ticket_words) {
        if (!fixed) PT_HIP(h, hipMemsetAsync(sc.tickets, 0, words * sizeof(uint32_t), h->stream));
        p.n_chunks = n_chunks;
        p.tickets = sc.tickets;
    }
    // Launches (ranges) of a few iterations: lanes have too few whole iterations to give away,
    // single paths change hands instead (measured at 1080p, demo scene: 1 iteration per
    // launch +23 %, 2: +18 %, 4: +7 %, 8: -3 %).
    const uint32_t per_range = div_up(n_iterations, p.n_chunks > 1 ? p.n_chunks : 1u);
    p.steal_unit = h->steal_unit ? h->steal_unit : (per_range <= 5 ? 1 : 4);
    return PT_OK;
}

// Enqueue n iterations of the megakernel on the handle's stream.
static int enqueue_megakernel(pt_context *h, float *dbuf, uint32_t n_iterations, const StealScratch *fixed)
{
    RenderParams p;
    fill_params(h, p);
    p.x0 = 0; p.y0 = 0; p.x1 = p.cam.width; p.y1 = p.cam.height;
    p.first_iteration = h->iteration + 1;
    p.n_iterations = n_iterations;
    p.accum_mode = h->accum_mode;
    p.overwrite = h->accum_mode == PT_ACCUM_RUNNING_MEAN ? (p.first_iteration == 1) : (h->fresh ? 1 : 0);
    p.buffer = dbuf;
    p.ray_counter = h->d_ray_counter;

    uint32_t grid_y;
    if (h->band_world > 1) {
        const uint32_t n_bands = div_up(p.cam.height, h->band_rows);
        const uint32_t owned = n_bands > h->band_rank ? div_up(n_bands - h->band_rank, h->band_world) : 0;
        p.band_rows = h->band_rows; p.band_rank = h->band_rank; p.band_world = h->band_world;
        grid_y = owned * (h->band_rows / 16);
    } else {
        p.band_rows = div_up(p.cam.height, 16) * 16; p.band_rank = 0; p.band_world = 1;
        grid_y = p.band_rows / 16;
    }
    const bool tiled = scene_streams(p.scene);
    if (grid_y > 0) {
        const uint32_t grid_x = div_up(p.cam.width, 16);
        const int rc = setup_steal(h, p, grid_x, grid_y, n_iterations, fixed);
        if (rc != PT_OK) return rc;
        PT_HIP(h, hipEventRecord(h->ev_start, h->stream));
        if (h->integrator == PT_INTEGRATOR_NEE_MIS)
            PT_HIP(h, launch_render_nee(p, grid_x, grid_y, tiled, false, h->stream));
        else
            PT_HIP(h, launch_render(p, grid_x, grid_y, tiled, false, h->stream));
        PT_HIP(h, hipEventRecord(h->ev_stop, h->stream));
        h->timed = true;
        h->last_launch = p.steal_scratch ? PT_LAUNCH_MEGAKERNEL_STEALING : PT_LAUNCH_MEGAKERNEL;
    }
    h->iteration += n_iterations;
    h->fresh = false;
    return PT_OK;
}

// Scenes with triangles go through the path pool or not at all.
static int triangles_renderable(pt_context *h, uint64_t last_iteration, uint32_t n_iterations)
{
    if (h->integrator != PT_INTEGRATOR_PATH)
        return fail(h, PT_ERR_STATE, "scenes with triangles: the NEE integrator samples sphere lights only (primitive.h:50-72)");
    if (h->wavefront || !takes_pool(h, last_iteration, n_iterations))
        return fail(h, PT_ERR_STATE, "scenes with triangles are rendered by the path pool only: PT_VARIANT_AUTO, "
                                     "PT_TUNE_POOL_ITERATIONS at its default, frames of fewer than 2^20 rows");
    return PT_OK;
}

// Enqueue n iterations on the handle's stream (async_call: see enqueue_wavefront).
static int enqueue(pt_context *h, float *dbuf, uint32_t n_iterations, bool async_call = false)
{
    if (!h->scene_ready) return fail(h, PT_ERR_STATE, "render before pt_init_scene");
    if (!dbuf) return fail(h, PT_ERR_INVALID, "render: buffer is NULL");
    if (n_iterations == 0) return PT_OK;
    if (h->iteration > 0xffffffffu - n_iterations) return fail(h, PT_ERR_INVALID, "iteration counter overflow");
    if (h->wavefront && h->integrator != PT_INTEGRATOR_PATH)
        return fail(h, PT_ERR_STATE, "the wavefront variant only implements PT_INTEGRATOR_PATH");
    if (h->wavefront && !wavefront_indexing_ok(h))
        return fail(h, PT_ERR_INVALID, "the wavefront variant takes frames of fewer than 2^20 rows and band_rows * world below 2^30");
    if (h->n_tris) {   // triangles: the path pool's kernels only
        const int rc = triangles_renderable(h, h->iteration + (uint64_t)n_iterations, n_iterations);
        if (rc != PT_OK) return rc;
        return enqueue_wavefront(h, dbuf, n_iterations, async_call);
    }
    if (takes_wavefront(h, h->iteration + (uint64_t)n_iterations, n_iterations)) {
        const int rc = enqueue_wavefront(h, dbuf, n_iterations, async_call);
        // PT_VARIANT_AUTO: when the queues cannot be had the megakernel renders the same image
        if (rc != PT_ERR_NOMEM || h->wavefront) return rc;
    }
    if (h->async_series) PT_HIP(h, join_async(h));
    return enqueue_megakernel(h, dbuf, n_iterations, nullptr);
}

int pt_run_n(pt_handle h, float *device_buffer, uint32_t n_iterations, int sync)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    int rc = enqueue(h, device_buffer, n_iterations);
    if (rc != PT_OK) return rc;
    if (sync) PT_HIP(h, hipStreamSynchronize(h->stream));
    return PT_OK;
}

int pt_run_async(pt_handle h, float *device_buffer, uint32_t n_iterations)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    return enqueue(h, device_buffer, n_iterations, true);
}

int pt_wait(pt_handle h)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    const bool series = h->async_series;
    PT_HIP(h, join_async(h));
    if (series) PT_HIP(h, hipEventRecord(h->ev_stop, h->stream));   // pt_last_kernel_ms: the whole series
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return PT_OK;
}

int pt_run_n_noalloc(pt_handle h, float *device_buffer, uint32_t n_iterations, int sync)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (h->async_series) PT_HIP(h, join_async(h));
    if (!h->scene_ready) return fail(h, PT_ERR_STATE, "render before pt_init_scene");
    if (!device_buffer) return fail(h, PT_ERR_INVALID, "render: buffer is NULL");
    if (h->wavefront)
        return fail(h, PT_ERR_STATE, "pt_run_n_noalloc: the wavefront variant needs queues that grow with the frame; "
                                     "select PT_VARIANT_AUTO or a megakernel variant");
    if (h->n_tris)
        return fail(h, PT_ERR_STATE, "pt_run_n_noalloc: scenes with triangles are rendered by the path pool, whose buffers grow with the frame");
    const uint64_t tiles = (uint64_t)div_up(h->camera.width, 16) * div_up(h->camera.height, 16);
    if (tiles + 1 > h->fixed_scratch.ticket_words)
        return fail(h, PT_ERR_INVALID, "pt_run_n_noalloc: %ux%u is %llu tiles of 16x16 pixels, the static scratch "
                    "holds tickets for %u", h->camera.width, h->camera.height, (unsigned long long)tiles,
                    h->fixed_scratch.ticket_words - 1);
    if (n_iterations == 0) return PT_OK;
    if (h->iteration > 0xffffffffu - n_iterations) return fail(h, PT_ERR_INVALID, "iteration counter overflow");
    const int rc = enqueue_megakernel(h, device_buffer, n_iterations, &h->fixed_scratch);
    if (rc != PT_OK) return rc;
    if (sync) PT_HIP(h, hipStreamSynchronize(h->stream));
    return PT_OK;
}

int pt_run(pt_handle h, float *buffer, int buffer_is_device)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (!buffer) return fail(h, PT_ERR_INVALID, "pt_run: buffer is NULL");
    if (buffer_is_device) return pt_run_n(h, buffer, 1, 1);

    const size_t bytes = (size_t)h->camera.width * h->camera.height * 3 * sizeof(float);
    if (!h->d_internal_buffer) {
        PT_HIP(h, hipMalloc((void **)&h->d_internal_buffer, bytes));
        PT_HIP(h, hipMemsetAsync(h->d_internal_buffer, 0, bytes, h->stream));
    }
    int rc = enqueue(h, h->d_internal_buffer, 1);
    if (rc != PT_OK) return rc;
    PT_HIP(h, hipMemcpyAsync(buffer, h->d_internal_buffer, bytes, hipMemcpyDeviceToHost, h->stream));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return PT_OK;
}

int pt_read_image(pt_handle h, const float *device_buffer, float *host_rgb)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (!device_buffer || !host_rgb) return fail(h, PT_ERR_INVALID, "pt_read_image: NULL pointer");
    if (h->async_series) PT_HIP(h, join_async(h));
    const size_t bytes = (size_t)h->camera.width * h->camera.height * 3 * sizeof(float);
    PT_HIP(h, hipMemcpyAsync(host_rgb, device_buffer, bytes, hipMemcpyDeviceToHost, h->stream));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    return PT_OK;
}

// ---- checkpoint / resume ---------------------------------------------------------------------
// File format and checks: pt_checkpoint.cpp (host code).  Here: the instance's side of it.
static CheckpointMeta checkpoint_meta(const pt_context *h)
{
    CheckpointMeta m;
    m.width = h->camera.width; m.height = h->camera.height;
    m.iteration = h->iteration;
    m.max_bounces = h->max_bounces;
    m.seed = h->seed;
    m.accum_mode = h->accum_mode;
    m.integrator = h->integrator;
    m.scene_hash = h->scene_hash;
    m.camera_hash = fingerprint(&h->camera, sizeof h->camera);
    return m;
}

int pt_save_checkpoint(pt_handle h, const float *device_buffer, const char *path)
{
    if (!h) return PT_ERR_INVALID;
    if (!device_buffer || !path) return fail(h, PT_ERR_INVALID, "pt_save_checkpoint: NULL argument");
    if (!h->scene_ready) return fail(h, PT_ERR_STATE, "pt_save_checkpoint before pt_init_scene");
    const size_t n = (size_t)h->camera.width * h->camera.height * 3;
    std::vector<float> img(n);
    const int rc = pt_read_image(h, device_buffer, img.data());
    if (rc != PT_OK) return rc;
    std::string err;
    if (checkpoint_write(path, checkpoint_meta(h), img.data(), &err) != kCkpOk)
        return fail(h, PT_ERR_INVALID, "pt_save_checkpoint: %s", err.c_str());
    return PT_OK;
}

int pt_load_checkpoint(pt_handle h, float *device_buffer, const char *path)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (!device_buffer || !path) return fail(h, PT_ERR_INVALID, "pt_load_checkpoint: NULL argument");
    if (!h->scene_ready) return fail(h, PT_ERR_STATE, "pt_load_checkpoint before pt_init_scene");
    std::vector<float> img;
    uint32_t iteration = 0;
    std::string err;
    const int rc = checkpoint_read(path, checkpoint_meta(h), &img, &iteration, &err);
    if (rc != kCkpOk) return fail(h, rc == kCkpMismatch ? PT_ERR_STATE : PT_ERR_INVALID, "pt_load_checkpoint: %s", err.c_str());
    if (h->async_series) PT_HIP(h, join_async(h));
    PT_HIP(h, hipMemcpyAsync(device_buffer, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    h->iteration = iteration;
    h->fresh = iteration == 0;
    return PT_OK;
}

int pt_ray_count(pt_handle h, uint64_t *out)
{
    if (!h || !out) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (h->async_series) PT_HIP(h, join_async(h));
    unsigned long long v = 0;
    PT_HIP(h, hipMemcpyAsync(&v, h->d_ray_counter, sizeof v, hipMemcpyDeviceToHost, h->stream));
    PT_HIP(h, hipStreamSynchronize(h->stream));
    *out = (uint64_t)v;
    return PT_OK;
}

int pt_reset_counters(pt_handle h)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (h->async_series) PT_HIP(h, join_async(h));
    PT_HIP(h, hipMemsetAsync(h->d_ray_counter, 0, sizeof(unsigned long long), h->stream));
    return PT_OK;
}

int pt_last_launch(pt_handle h, int *kind)
{
    if (!h || !kind) return PT_ERR_INVALID;
    *kind = h->last_launch;
    return PT_OK;
}

int pt_last_kernel_ms(pt_handle h, float *ms)
{
    if (!h || !ms) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (!h->timed) return fail(h, PT_ERR_STATE, "pt_last_kernel_ms: nothing was launched yet");
    if (h->async_series) {   // an open series of pt_run_async calls: its time so far
        PT_HIP(h, join_async(h));
        PT_HIP(h, hipEventRecord(h->ev_stop, h->stream));
    }
    PT_HIP(h, hipEventSynchronize(h->ev_stop));
    PT_HIP(h, hipEventElapsedTime(ms, h->ev_start, h->ev_stop));
    return PT_OK;
}

int pt_debug_trace(pt_handle h, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                   uint32_t iteration, int32_t *hit_ids, float *radiance, float *pixel_rgb,
                   uint64_t *ray_count)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (!h->scene_ready) return fail(h, PT_ERR_STATE, "pt_debug_trace before pt_init_scene");
    if (x0 >= x1 || y0 >= y1 || x1 > h->camera.width || y1 > h->camera.height)
        return fail(h, PT_ERR_INVALID, "pt_debug_trace: bad rectangle [%u,%u)x[%u,%u)", x0, x1, y0, y1);
    if (iteration == 0) return fail(h, PT_ERR_INVALID, "pt_debug_trace: iterations are 1-based");
    if (h->n_tris) {
        const int trc = triangles_renderable(h, iteration, 1);
        if (trc != PT_OK) return trc;
    }
    if (h->async_series) PT_HIP(h, join_async(h));

    const size_t n_pix = (size_t)(x1 - x0) * (y1 - y0);
    const size_t n_paths = n_pix * 4;
    const size_t hit_elems = n_paths * (size_t)(h->max_bounces + 1);
    int32_t *d_hits = nullptr;
    float *d_rad = nullptr, *d_rgb = nullptr;
    unsigned long long *d_cnt = nullptr;
    int rc = PT_OK;
    hipError_t e = hipSuccess;
    std::vector<int32_t> init;

    #define DBG_HIP(call) do { e = (call); if (e != hipSuccess) { rc = fail(h, PT_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e)); goto cleanup; } } while (0)
    if (hit_ids) {
        init.assign(hit_elems, PT_HIT_NOT_TRACED);
        DBG_HIP(hipMalloc((void **)&d_hits, hit_elems * sizeof(int32_t)));
        DBG_HIP(hipMemcpyAsync(d_hits, init.data(), hit_elems * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    }
    if (radiance) DBG_HIP(hipMalloc((void **)&d_rad, n_paths * 3 * sizeof(float)));
    if (pixel_rgb) DBG_HIP(hipMalloc((void **)&d_rgb, n_pix * 3 * sizeof(float)));
    DBG_HIP(hipMalloc((void **)&d_cnt, sizeof(unsigned long long)));
    DBG_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), h->stream));
    // The rectangle goes through the kernels a render of this instance would use: the per-bounce
    // wavefront launches, the work-stealing megakernel, or the plain one.
    if (takes_wavefront(h, iteration, 1) && (h->integrator == PT_INTEGRATOR_PATH || !h->wavefront) && iteration < (1u << 30) &&
        wavefront_indexing_ok(h)) {
        const int mode = wavefront_mode(h, 1);
        size_t counts_bytes = 0;
        rc = ensure_wavefront(h, mode, n_paths, wavefront_sub_capacity(x1 - x0, y1 - y0, 1), 1, &counts_bytes);
        if (rc != PT_OK) goto cleanup;
        WavefrontParams p;
        fill_wavefront(h, p);
        p.x0 = x0; p.y0 = y0; p.rect_w = x1 - x0; p.rect_h = y1 - y0;
        p.first_iteration = iteration;
        p.n_iterations = 1;
        p.accum_mode = PT_ACCUM_SUM;
        p.overwrite = 1;
        p.buffer = nullptr;
        p.ray_counter = d_cnt;
        p.hit_ids = d_hits; p.pixel_rgb = d_rgb;
        p.store_all_paths = 1;   // `radiance` is read for every path
        rc = launch_wavefront_batch(h, p, mode, counts_bytes, 0, h->stream, nullptr, true, radiance && mode != 3);
        if (rc != PT_OK) goto cleanup;
        // the batch's per-path radiance buffer has the layout of `radiance`
        if (radiance) DBG_HIP(hipMemcpyAsync(radiance, h->wf_pipe[0].radiance, n_paths * 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    } else {
        RenderParams p;
        fill_params(h, p);
        p.x0 = x0; p.y0 = y0; p.x1 = x1; p.y1 = y1;
        p.band_rows = div_up(y1 - y0, 16) * 16; p.band_rank = 0; p.band_world = 1;
        p.first_iteration = iteration;
        p.n_iterations = 1;
        p.accum_mode = PT_ACCUM_SUM;
        p.overwrite = 1;
        p.buffer = nullptr;
        p.ray_counter = d_cnt;
        p.hit_ids = d_hits; p.path_rad = d_rad; p.pixel_rgb = d_rgb;
        const bool tiled = scene_streams(p.scene);
        rc = setup_steal(h, p, div_up(x1 - x0, 16), p.band_rows / 16, 1, nullptr);
        if (rc != PT_OK) goto cleanup;
        if (h->integrator == PT_INTEGRATOR_NEE_MIS)
            DBG_HIP(launch_render_nee(p, div_up(x1 - x0, 16), p.band_rows / 16, tiled, true, h->stream));
        else
            DBG_HIP(launch_render(p, div_up(x1 - x0, 16), p.band_rows / 16, tiled, true, h->stream));
        if (radiance) DBG_HIP(hipMemcpyAsync(radiance, d_rad, n_paths * 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    }
    if (hit_ids) DBG_HIP(hipMemcpyAsync(hit_ids, d_hits, hit_elems * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    if (pixel_rgb) DBG_HIP(hipMemcpyAsync(pixel_rgb, d_rgb, n_pix * 3 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    {
        unsigned long long cnt = 0;
        DBG_HIP(hipMemcpyAsync(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost, h->stream));
        DBG_HIP(hipStreamSynchronize(h->stream));
        if (ray_count) *ray_count = (uint64_t)cnt;
    }
    #undef DBG_HIP
cleanup:
    if (rc != PT_OK) (void)hipStreamSynchronize(h->stream);
    if (d_hits) (void)hipFree(d_hits);
    if (d_rad) (void)hipFree(d_rad);
    if (d_rgb) (void)hipFree(d_rgb);
    if (d_cnt) (void)hipFree(d_cnt);
    return rc;
}

int pt_selftest_eval(pt_handle h, int op, const float *a, const float *b, float *out, uint64_t n)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (op < 0 || op > 4 || !a || !out || (op == 2 && !b)) return fail(h, PT_ERR_INVALID, "pt_selftest_eval: bad arguments");
    if (n == 0) return PT_OK;
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    int rc = PT_OK;
    hipError_t e = hipSuccess;
    #define ST_HIP(call) do { e = (call); if (e != hipSuccess) { rc = fail(h, PT_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(e)); goto done; } } while (0)
    ST_HIP(hipMalloc((void **)&da, n * sizeof(float)));
    ST_HIP(hipMalloc((void **)&dout, n * sizeof(float)));
    ST_HIP(hipMemcpyAsync(da, a, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (b) {
        ST_HIP(hipMalloc((void **)&db, n * sizeof(float)));
        ST_HIP(hipMemcpyAsync(db, b, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
    ST_HIP(launch_selftest_eval(op, da, db, dout, n, h->stream));
    ST_HIP(hipMemcpyAsync(out, dout, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    ST_HIP(hipStreamSynchronize(h->stream));
done:
    if (rc != PT_OK) (void)hipStreamSynchronize(h->stream);
    if (da) (void)hipFree(da);
    if (db) (void)hipFree(db);
    if (dout) (void)hipFree(dout);
    return rc;
}

int pt_selftest_sample_material(pt_handle h, const pt_material *material, const float *in, float *out, uint64_t n)
{
    if (!h) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (!material || !in || !out) return fail(h, PT_ERR_INVALID, "pt_selftest_sample_material: bad arguments");
    if (n == 0) return PT_OK;
    const float4 cc = material_table_entry(*material);
    float *din = nullptr, *dout = nullptr;
    int rc = PT_OK;
    hipError_t e = hipSuccess;
    ST_HIP(hipMalloc((void **)&din, n * 8 * sizeof(float)));
    ST_HIP(hipMalloc((void **)&dout, n * 8 * sizeof(float)));
    ST_HIP(hipMemcpyAsync(din, in, n * 8 * sizeof(float), hipMemcpyHostToDevice, h->stream));
    ST_HIP(launch_selftest_material(material->type, cc, 1.f / material->coef, din, dout, n, h->stream));
    ST_HIP(hipMemcpyAsync(out, dout, n * 8 * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    ST_HIP(hipStreamSynchronize(h->stream));
done:
    if (rc != PT_OK) (void)hipStreamSynchronize(h->stream);
    if (din) (void)hipFree(din);
    if (dout) (void)hipFree(dout);
    return rc;
}

int pt_selftest_exhaustive(pt_handle h, int op, uint64_t *mismatches)
{
    if (!h || !mismatches) return PT_ERR_INVALID;
    DeviceGuard on_device(h->device);
    if (op < 0 || op > 3) return fail(h, PT_ERR_INVALID, "pt_selftest_exhaustive: op must be 0, 1, 2 or 3");
    unsigned long long *d = nullptr, v = 0;
    int rc = PT_OK;
    hipError_t e = hipSuccess;
    ST_HIP(hipMalloc((void **)&d, sizeof v));
    ST_HIP(hipMemsetAsync(d, 0, sizeof v, h->stream));
    ST_HIP(launch_selftest_exhaustive(op, d, h->stream));
    ST_HIP(hipMemcpyAsync(&v, d, sizeof v, hipMemcpyDeviceToHost, h->stream));
    ST_HIP(hipStreamSynchronize(h->stream));
    *mismatches = (uint64_t)v;
    #undef ST_HIP
done:
    if (rc != PT_OK) (void)hipStreamSynchronize(h->stream);
    if (d) (void)hipFree(d);
    return rc;
}

} // extern "C"
