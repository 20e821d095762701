// Launch interface between the C-ABI host code (pt_api.cpp) and the HIP kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace ptmi {

// Spheres per LDS tile of the streaming (large-scene) kernel: 1024 x 16 B = 16 KiB.
constexpr uint32_t kTileSpheres = 1024;
// Scenes up to this many spheres are staged in LDS once per workgroup
// (32 B per sphere -> 16 KiB at the limit).
constexpr uint32_t kMaxResidentSpheres = 512;

// Passed by value as the kernel argument (wave-uniform, lives in SGPRs).
struct RenderParams {
    // camera: what Camera::GenerateRay reads (camera.h:66-72)
    float cam_pos[3], cam_pxx[3], cam_pxy[3], cam_first[3];
    uint32_t width, height;
    // pixel rectangle [x0,x1) x [y0,y1) and the row-band ownership inside it
    uint32_t x0, y0, x1, y1;
    uint32_t band_rows, band_rank, band_world;
    // scene, SoA in device memory
    const float4 *sph;   // (cx, cy, cz, radius*radius)
    const float4 *aux;   // (1/radius, materialId bits, lightId bits, radius)
    const float4 *mat;   // (r, g, b, coef)
    const int *mat_type; // material.h:25
    const float4 *light; // (r, g, b, -) intensity
    uint32_t n_prims, n_mats, n_lights;
    // sampling
    uint64_t seed;
    int max_bounces;
    uint32_t first_iteration; // 1-based number of the first iteration rendered
    uint32_t n_iterations;
    int accum_mode;           // 0 running mean, 1 sum
    int overwrite;            // 1: start from zero instead of reading the buffer
    int regen_min_idle;       // 1..64, idle lanes needed before a wave starts new paths
    // outputs
    float *buffer;                     // width*height*3, may be NULL (debug trace)
    unsigned long long *ray_counter;   // += number of Intersect calls
    int32_t *hit_ids;                  // debug: [rect pixel][4][max_bounces+1]
    float *path_rad;                   // debug: [rect pixel][4][3]
    float *pixel_rgb;                  // debug: [rect pixel][3]
};

size_t render_lds_bytes(const RenderParams &p, bool tiled);

// grid_x x grid_y workgroups of 256 threads, each a 16x16 pixel tile.
hipError_t launch_render(const RenderParams &p, uint32_t grid_x, uint32_t grid_y, bool tiled,
                         bool record, hipStream_t stream);

// Arithmetic self-tests (device pointers): out[i] = op(a[i], b[i]) with the device functions the
// trace loop uses; exhaustive comparison of the short sqrt / reciprocal / division sequences
// against the compiler's IEEE expansions.
hipError_t launch_selftest_eval(int op, const float *a, const float *b, float *out, uint64_t n, hipStream_t stream);
hipError_t launch_selftest_exhaustive(int op, unsigned long long *mismatches, hipStream_t stream);

} // namespace ptmi
