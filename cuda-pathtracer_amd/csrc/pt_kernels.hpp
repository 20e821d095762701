// Launch interface between the C-ABI host code (pt_api.cpp) and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace ptmi {

// Spheres per LDS tile of the streaming (large-scene) kernels: 1024 x 16 B = 16 KiB.
constexpr uint32_t kTileSpheres = 1024;
// Scenes up to this many spheres are staged in LDS once per workgroup
// (32 B per sphere -> 16 KiB at the limit).
constexpr uint32_t kMaxResidentSpheres = 512;
// Triangles per LDS tile of the streaming kernels (three float4 each: 12 KiB), and the most a resident scene holds.
constexpr uint32_t kTileTriangles = 256;
constexpr uint32_t kMaxResidentTriangles = 256;
// ... and only while spheres, materials and lights together stay below this many bytes of LDS
// (a scene may carry far more materials than spheres).
constexpr size_t kMaxResidentBytes = 24 * 1024;
// Scenes with more spheres than this get a sphere BVH (pt_bvh.cpp).  Measured at 1080p on clouds of
// small spheres in the demo room (tools/bvh_crossover.py): the scan from LDS wins up to ~110
// spheres (2.5 vs 2.9 ms per iteration), the BVH from ~170 (3.3 vs 3.4 ms; 3.5 vs 4.8 ms at 260).
constexpr uint32_t kBvhMinSpheres = 192;
// Scenes up to this many spheres also keep a per-lane ring of pre-generated camera rays in LDS.
constexpr uint32_t kMaxRingSpheres = 24;
// Work stealing inside the wave: iterations per pixel and launch that other lanes may take over.
constexpr uint32_t kStealSlots = 8;
// Scratch of the work-stealing megakernel.  A stolen unit's result only lives until the wave that
// produced it ends, so the scratch is a pool of per-workgroup blocks, sized by the workgroups the
// chip can hold at once and not by the frame: kStealPoolXcc sub-pools (one per XCD, so that a block
// is only ever touched through one L2) of kStealPoolPerXcc blocks (32 CUs x 8 workgroups of four
// waves), each [kStealSlots][256 lanes][3] floats.  A workgroup takes a free block of its XCD's
// sub-pool when it starts (one bit per block in `mask`) and returns it when it ends; a workgroup
// that finds none renders without stealing.
constexpr uint32_t kStealPoolXcc = 8;
constexpr uint32_t kStealPoolPerXcc = 256;
constexpr uint32_t kStealBlockFloats = kStealSlots * 256 * 3;
constexpr size_t kStealPoolBytes = (size_t)kStealPoolXcc * kStealPoolPerXcc * kStealBlockFloats * sizeof(float);   // 48 MiB
// occupancy bits of the pool: per XCD one 128-byte line, of which the first 256 bits are used
constexpr uint32_t kStealMaskWordsPerXcc = 16;
constexpr size_t kStealMaskBytes = (size_t)kStealPoolXcc * kStealMaskWordsPerXcc * sizeof(unsigned long long);     // 1 KiB

// What Camera::GenerateRay reads (camera.h:66-72).
struct CameraParams {
    float pos[3], pxx[3], pxy[3], first[3];
    uint32_t width, height;
};

// Where a launch of the work-stealing megakernel finds its scratch.
struct StealScratch {
    float *pool;                  // kStealPoolBytes
    unsigned long long *mask;     // kStealMaskBytes
    uint32_t *tickets;            // iteration ranges: start-order counter + one word per tile (may be NULL)
    uint32_t ticket_words;
};

// The statically allocated scratch (a file-scope __device__ array in pt_kernels.hip): pool, occupancy
// bits and kStaticTicketWords ticket words, contiguous in this order.
constexpr uint32_t kStaticTicketWords = 1u << 16;
constexpr size_t kStaticScratchBytes = kStealPoolBytes + kStealMaskBytes + (size_t)kStaticTicketWords * sizeof(uint32_t);
// Address of the CURRENT device's copy (`device` must be the current device), looked up with
// hipGetSymbolAddress once per device and kept.
hipError_t static_steal_scratch(int device, StealScratch *out);

// Sphere BVH in device memory (layout: pt_bvh.hpp); n_nodes == 0 when there is none.
struct BvhParams {
    const uint4 *nodes;       // four uint4 per four-wide inner node (16-bit grid planes + child references), node 0 is the root
    const float4 *leaf_sph;   // [leaf][4]
    const int4 *leaf_index;   // [leaf]
    const float4 *big_sph;
    const int *big_index;
    uint32_t n_nodes, n_big, n_leaves;
    uint32_t depth;           // entries of the per-lane traversal stack
    uint32_t budget;          // box/leaf visits after which a ray is finished by a wave-wide scan
    uint32_t node_steps;      // inner-node steps a wave takes per leaf step (the lanes of a wave alternate in lockstep)
    float centre[3];
    float half_diag, r_min;
    float grid_origin[3], grid_step[3];   // plane = grid_origin + q * grid_step
    unsigned long long *stats;   // 32 counters, written only by -DPT_BVH_STATS builds
};

// Scene, SoA in device memory.
struct SceneParams {
    BvhParams bvh;
    const float4 *sph;   // (cx, cy, cz, radius*radius)
    const float4 *aux;   // (1/radius, materialId bits, lightId bits, radius)
    const float4 *mat;   // (r, g, b, coef)
    const int2 *mat_type; // (material.h:25, bits of 1.f / coef)
    const float4 *light; // (r, g, b, -) intensity
    const float4 *light_geom; // point light: (x, y, z, bits 0); area light: (bits sphere index, -, -, bits 1)
    uint32_t n_prims, n_mats, n_lights;
    // Triangles (path pool only): four float4 per triangle -- (v0, material id bits), (v1 - v0, light id bits),
    // (v2 - v0, -), (unit normal, -); triangle k is primitive n_prims + k
    const float4 *tri;
    uint32_t n_tris;
};

// Megakernel argument, passed by value (wave-uniform, lives in SGPRs).
struct RenderParams {
    CameraParams cam;
    SceneParams scene;
    // pixel rectangle [x0,x1) x [y0,y1) and the row-band ownership inside it
    uint32_t x0, y0, x1, y1;
    uint32_t band_rows, band_rank, band_world;
    // sampling
    uint64_t seed;
    int max_bounces;
    uint32_t first_iteration; // 1-based number of the first iteration rendered
    uint32_t n_iterations;
    int accum_mode;           // 0 running mean, 1 sum
    int overwrite;            // 1: start from zero instead of reading the buffer
    int regen_min_idle;       // 1..64, idle lanes needed before a wave starts new paths
    int small_scene_opts;     // bit 0: camera-ray ring (scenes <= kMaxRingSpheres)
    // work stealing inside the wave (render_steal_kernel): the pool of per-workgroup scratch blocks
    // (kStealPoolBytes) for the units rendered by another lane than the pixel's, and its occupancy
    // bits (kStealMaskBytes, all zero between launches); NULL: no stealing
    float *steal_scratch;
    unsigned long long *steal_mask;
    int steal_unit;           // paths that change hands at a time: 4 (a pixel-iteration) or 1
    // chunked launch of render_steal_kernel: n_chunks consecutive iteration ranges, one workgroup per
    // tile and range; tickets[0] = start-order counter, tickets[1 + tile] = ranges of the tile done
    uint32_t n_chunks;
    uint32_t *tickets;
    // outputs
    float *buffer;                     // width*height*3, may be NULL (debug trace)
    unsigned long long *ray_counter;   // += number of Intersect calls
    int32_t *hit_ids;                  // debug: [rect pixel][4][max_bounces+1]
    float *path_rad;                   // debug: [rect pixel][4][3]
    float *pixel_rgb;                  // debug: [rect pixel][3]
};

size_t scene_lds_bytes(const SceneParams &s, bool tiled);
// LDS of the BVH kernels: the per-lane traversal stacks, [depth][256] entries of 8 bytes.
inline size_t bvh_stack_lds_bytes(const BvhParams &b) { return (size_t)(b.depth ? b.depth : 1) * 256 * 8; }

// grid_x x grid_y workgroups of 256 threads, each a 16x16 pixel tile.
hipError_t launch_render(const RenderParams &p, uint32_t grid_x, uint32_t grid_y, bool tiled,
                         bool record, hipStream_t stream);

hipError_t read_lane_stats(unsigned long long out[64]);
hipError_t add_lane_stats_wavefront(unsigned long long out[64]);
hipError_t add_lane_stats_nee(unsigned long long out[64]);

hipError_t launch_render_steal(const RenderParams &p, uint32_t grid_x, uint32_t grid_y, bool tiled, bool nee,
                               bool record, hipStream_t stream);

// The opt-in NEE + MIS integrator (pt_nee.hip); same launch geometry.
hipError_t launch_render_nee(const RenderParams &p, uint32_t grid_x, uint32_t grid_y, bool tiled, bool record,
                             hipStream_t stream);

// ---- wavefront (path-compaction) variant ------------------------------------------------------
// A batch holds n_iterations x width x height x 4 paths.  Every bounce is one kernel launch over
// a compacted queue of live paths in device memory; the emission a path collects is added to its
// entry of `path_radiance` bounce by bounce, in the order the reference adds it to L, and
// wf_resolve folds the entries into the frame buffer in the reference's order.
constexpr uint32_t kWfSubQueues = 64;     // independent sub-queues (one compaction counter each)
constexpr uint32_t kWfCountStride = 32;   // uint32 per counter: every counter on its own 128-B line

// Path record of the queues, 44 bytes in three arrays.  The path's radiance is not carried: a
// bounce that meets a light adds its contribution to the path's entry of `path_radiance`.
struct WavefrontTail { float tz; uint32_t pix, it_sub; };   // (Tz, pixel index, iteration << 2 | sub-sample)
struct WavefrontQueue {   // [kWfSubQueues][sub_capacity]
    float4 *a;            // (ox, oy, oz, dx)
    float4 *b;            // (dy, dz, Tx, Ty)
    WavefrontTail *c;
};

struct WavefrontParams {
    CameraParams cam;
    SceneParams scene;
    uint64_t seed;
    int max_bounces;
    uint32_t x0, y0, rect_w, rect_h;   // pixel rectangle of the batch (a render: the whole frame)
    uint32_t band_rows, band_rank, band_world;   // band_world > 1: the rectangle's rows are the owned row bands
    uint32_t first_iteration;   // of the batch
    uint32_t n_iterations;      // in the batch
    int bounce;                 // bounce index handled by this launch
    WavefrontQueue in, out;     // in is unused for bounce 0 (camera rays are generated)
    int dynamic_fetch;          // BVH scenes: persistent waves refill when fewer lanes than this still
                                // traverse (0: static chunks of 256 paths)
    uint32_t sub_capacity;      // paths per sub-queue, a multiple of 256
    int whole_paths;            // BVH scenes with dynamic fetch: one launch traces whole paths (no queues)
    int path_pool;              // resident scenes: one launch of persistent waves that take camera rays from a pool (no queues)
    int nee;                    // PT_INTEGRATOR_NEE_MIS (path pool, or whole paths from the camera ray on): a lane's pass
                                // is a path ray or the shadow ray of the vertex it shaded in the pass before
    int pool_workgroups_per_cu; // path pool: persistent workgroups launched per CU (0: as many as fit, at most 8)
    uint32_t *counts;           // [max_bounces + 2][kWfSubQueues][kWfCountStride]: word 0 = paths alive
                                // in sub-queue q entering bounce k, word 1 = fetch cursor of the
                                // dynamic-fetch kernel over those paths
    float *path_radiance;       // [paths][3], never zeroed: a path that collects light has an entry (the queue-fed
                                // kernels add to it bounce by bounce, the path pool stores it when the path
                                // ends) and is marked in path_mask
    uint32_t *path_mask;        // one byte per pixel-iteration of the batch, bit s = sub-sample s has an entry in
                                // path_radiance; zero when the batch starts, left at zero by the resolve
    int store_all_paths;        // path pool, parity hook: black paths are stored and marked too
    // resolve
    int accum_mode, overwrite;
    float *buffer;              // may be NULL (parity hook)
    unsigned long long *ray_counter;
    // parity hook (NULL in renders): [path of the batch][max_bounces + 1] hit indices, [pixel][3] colours
    int32_t *hit_ids;
    float *pixel_rgb;
};

hipError_t launch_wavefront_step(const WavefrontParams &p, bool tiled, hipStream_t stream);
hipError_t launch_wavefront_resolve(const WavefrontParams &p, hipStream_t stream);

// Arithmetic self-tests (device pointers): out[i] = op(a[i], b[i]) with the device functions the
// trace loop uses; exhaustive comparison of the short sqrt / reciprocal / division sequences
// against the compiler's IEEE expansions.
hipError_t launch_selftest_eval(int op, const float *a, const float *b, float *out, uint64_t n, hipStream_t stream);
hipError_t launch_selftest_material(int type, float4 color_coef, float inv_coef, const float *in, float *out, uint64_t n, hipStream_t stream);
hipError_t launch_selftest_exhaustive(int op, unsigned long long *mismatches, hipStream_t stream);

} // namespace ptmi
