// Render kernels for gfx950 (MI355X, wave64).
//
// One launch renders n_iterations iterations of the reference's frame
// (GenerateRayPool + RenderKernel + Trace, pathtracer.cu:62-170) for a pixel rectangle.
//
// Mapping: a 256-thread workgroup owns a 16x16 pixel tile (as the reference's launch,
// pathtracer.cu:227-230), each wave an 8x8 sub-tile, each lane ONE pixel for the whole
// launch.  A lane walks its pixel's paths in the reference's order -- iteration by
// iteration, sub-sample 0..3 -- so the pixel sum ((0+L0)+L1)+L2)+L3 and the running mean are
// evaluated in exactly the reference's order while the accumulator stays in registers.
//
// "Persistent lanes with path regeneration": the loop body is one Intersect plus the shading
// of one bounce; a lane whose path ended starts its next path at the top of the body instead
// of idling until the longest path of the wave is finished.  No ray pool and no random-number
// buffer exist: camera rays and Philox numbers are generated in registers.
//
// Scene data: spheres as float4 (cx, cy, cz, r*r) read as wave-uniform LDS broadcasts.  Small
// scenes are staged once per workgroup; larger ones stream through a 1024-sphere LDS tile.
#include "pt_kernels.hpp"
#include "pt_device.hpp"

namespace ptmi {

namespace {

constexpr int kBlockThreads = 256;

// Wave-wide sum of a 32-bit counter (wave64), result valid in every lane.
PT_DEV uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct LdsScene {
    const float4 *sph;   // [n_prims] or one tile
    const float4 *aux;   // (1/radius, materialId bits, lightId bits, radius)
    const float4 *mat;   // (r, g, b, coef)
    const int *mat_type;
    const float4 *light; // (r, g, b, -)
};

// Scan `count` spheres that sit in LDS starting at global index `base`.
PT_DEV void scan_lds(const float4 *sph, int base, int count, V3 o, V3 d, float &tmax, int &best)
{
    int i = 0;
    for (; i + 4 <= count; i += 4) {
        const float4 s0 = sph[i], s1 = sph[i + 1], s2 = sph[i + 2], s3 = sph[i + 3];
        sphere_test(s0, base + i, o, d, tmax, best);
        sphere_test(s1, base + i + 1, o, d, tmax, best);
        sphere_test(s2, base + i + 2, o, d, tmax, best);
        sphere_test(s3, base + i + 3, o, d, tmax, best);
    }
    for (; i < count; ++i) sphere_test(sph[i], base + i, o, d, tmax, best);
}

template <bool TILED, bool RECORD>
__global__ __launch_bounds__(kBlockThreads) void render_kernel(const RenderParams p)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;

    // ---- stage the scene -----------------------------------------------------------------
    LdsScene sc;
    float4 *lds_sph = reinterpret_cast<float4 *>(lds_raw);
    if (!TILED) {
        float4 *lds_aux = lds_sph + p.n_prims;
        float4 *lds_mat = lds_aux + p.n_prims;
        float4 *lds_light = lds_mat + p.n_mats;
        int *lds_type = reinterpret_cast<int *>(lds_light + p.n_lights);
        for (uint32_t i = tid; i < p.n_prims; i += kBlockThreads) {
            lds_sph[i] = p.sph[i];
            lds_aux[i] = p.aux[i];
        }
        for (uint32_t i = tid; i < p.n_mats; i += kBlockThreads) {
            lds_mat[i] = p.mat[i];
            lds_type[i] = p.mat_type[i];
        }
        for (uint32_t i = tid; i < p.n_lights; i += kBlockThreads) lds_light[i] = p.light[i];
        __syncthreads();
        sc.sph = lds_sph; sc.aux = lds_aux; sc.mat = lds_mat; sc.mat_type = lds_type; sc.light = lds_light;
    } else {
        sc.sph = lds_sph; sc.aux = p.aux; sc.mat = p.mat; sc.mat_type = p.mat_type; sc.light = p.light;
    }

    // ---- pixel owned by this lane ----------------------------------------------------------
    // blockIdx.y enumerates the 16-row block rows of the bands this instance owns.
    const uint32_t blocks_per_band = p.band_rows >> 4;
    const uint32_t band_local = blockIdx.y / blocks_per_band;
    const uint32_t band = band_local * p.band_world + p.band_rank;
    const uint32_t yb = p.y0 + band * p.band_rows + (blockIdx.y - band_local * blocks_per_band) * 16u;
    const uint32_t x = p.x0 + blockIdx.x * 16u + (uint32_t)((wave & 1) * 8 + (lane & 7));
    const uint32_t y = yb + (uint32_t)((wave >> 1) * 8 + (lane >> 3));
    const bool valid = x < p.x1 && y < p.y1;
    const uint32_t pix = y * p.width + x;
    const uint32_t rect_pix = (y - p.y0) * (p.x1 - p.x0) + (x - p.x0);

    const V3 cam_pos = mk(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
    const V3 cam_pxx = mk(p.cam_pxx[0], p.cam_pxx[1], p.cam_pxx[2]);
    const V3 cam_pxy = mk(p.cam_pxy[0], p.cam_pxy[1], p.cam_pxy[2]);
    const V3 cam_first = mk(p.cam_first[0], p.cam_first[1], p.cam_first[2]);
    const uint32_t k0 = (uint32_t)p.seed, k1 = (uint32_t)(p.seed >> 32);
    const int mb = p.max_bounces;

    // ---- lane state --------------------------------------------------------------------------
    uint32_t it = p.first_iteration;
    const uint32_t it_end = p.first_iteration + p.n_iterations;
    int sub = 0;
    bool done = !valid || p.n_iterations == 0;
    bool alive = false;
    V3 acc = mk(0.f, 0.f, 0.f);
    if (valid && !p.overwrite && p.buffer) {
        const float *px = p.buffer + (size_t)pix * 3;
        acc = mk(px[0], px[1], px[2]);
    }
    V3 c = mk(0.f, 0.f, 0.f);
    V3 o = cam_pos, d = mk(0.f, 0.f, 1.f), T = mk(1.f, 1.f, 1.f), L = mk(0.f, 0.f, 0.f);
    int b = 0;
    uint32_t nrays = 0;

    for (;;) {
        // ---- path regeneration ---------------------------------------------------------------
        // regen_min_idle = 1: any idle lane restarts at once; 64: lanes restart only when the
        // whole wave is idle (the classic one-path-per-lane-at-a-time megakernel).
        bool want = !done && !alive;
        if (p.regen_min_idle > 1) {
            const int idle = __popcll(__ballot(!alive));
            want = want && idle >= p.regen_min_idle;
        }
        if (want) {
            // CameraSamples (pathtracer.cu:34-48) + Camera::GenerateRay (camera.h:66-72)
            uint32_t r[4];
            philox4x32_10(pix, it, (uint32_t)(sub >> 1), STREAM_CAMERA, k0, k1, r);
            const float ju = u01((sub & 1) ? r[2] : r[0]) * ((sub & 1) ? 0.5f : -0.5f);
            const float jv = u01((sub & 1) ? r[3] : r[1]) * ((sub & 2) ? 0.5f : -0.5f);
            const float sx = (float)(int)x + ju;
            const float sy = (float)(int)y + jv;
            o = cam_pos;
            d = normalize((cam_first - cam_pxy * sy) + cam_pxx * sx);
            T = mk(1.f, 1.f, 1.f);
            L = mk(0.f, 0.f, 0.f);
            b = 0;
            alive = true;
        }
        if (TILED) {
            if (!__syncthreads_or(alive)) break;
        } else {
            if (!__any(alive)) break;
        }

        // ---- Intersect (scene.h:71-94) ---------------------------------------------------------
        float tmax = PT_INF;
        int best = -1;
        if (TILED) {
            for (uint32_t base = 0; base < p.n_prims; base += kTileSpheres) {
                const uint32_t count = min(kTileSpheres, p.n_prims - base);
                __syncthreads();
                for (uint32_t i = tid; i < count; i += kBlockThreads) lds_sph[i] = p.sph[base + i];
                __syncthreads();
                if (alive) scan_lds(lds_sph, (int)base, (int)count, o, d, tmax, best);
            }
        } else {
            if (alive) scan_lds(sc.sph, 0, (int)p.n_prims, o, d, tmax, best);
        }

        if (alive) {
            ++nrays;
            if (RECORD) {
                if (p.hit_ids)
                    p.hit_ids[((size_t)rect_pix * 4 + sub) * (size_t)(mb + 1) + b] = best;
            }
            bool end = true;
            if (best >= 0) {
                // hit point and normal: scene.h:86-87, primitive.h:74
                const float4 s = TILED ? p.sph[best] : sc.sph[best];
                const float4 a = sc.aux[best];
                const V3 P = o + d * tmax;
                const V3 n = (P - mk(s.x, s.y, s.z)) * a.x;
                const V3 wo = d;
                const uint32_t mat_id = __float_as_uint(a.y);
                const int light_id = __float_as_int(a.z);

                // emission on every light hit: pathtracer.cu:134-137, Light::L light.h:43-45
                if (light_id > -1) {
                    const float4 li = sc.light[light_id];
                    const V3 le = dot(n, -wo) > 0.f ? mk(li.x, li.y, li.z) : mk(0.f, 0.f, 0.f);
                    L = L + T * le;
                }

                uint32_t r[4];
                philox4x32_10(pix, it, (uint32_t)b, STREAM_PATH, k0, k1, r);
                const float4 m = sc.mat[mat_id];
                const int mtype = sc.mat_type[mat_id];
                V3 wi;
                float pdf;
                const V3 f = sample_material(mtype, mk(m.x, m.y, m.z), m.w, wo, wi, pdf, n,
                                             u01(r[0]), u01(r[1]));
                // pathtracer.cu:146-161
                if (!((f.x == 0.f && f.y == 0.f && f.z == 0.f) || pdf == 0.f)) {
                    T = T * vdiv(f * __builtin_fabsf(dot(wi, n)), pdf);
                    o = P;
                    d = wi;
                    end = false;
                    if (b > 3) {
                        const float gb = T.y > T.z ? T.y : T.z;   // Color::Max, geometry.h:474
                        const float mx = T.x > gb ? T.x : gb;
                        const float q = .5f < mx ? .5f : mx;      // min macro, globals.h:66
                        if (u01(r[2]) > q) end = true;
                        else T = vdiv(T, q);
                    }
                    if (b == mb) end = true;
                    ++b;
                }
            }
            if (end) {
                if (RECORD) {
                    if (p.path_rad) {
                        float *pr = p.path_rad + ((size_t)rect_pix * 4 + sub) * 3;
                        pr[0] = L.x; pr[1] = L.y; pr[2] = L.z;
                    }
                }
                c = c + L;
                alive = false;
                if (++sub == 4) {
                    // RenderKernel, pathtracer.cu:101-109
                    c = vdiv(c, 4.f);
                    if (p.accum_mode == 0) {
                        const float inv_it = 1.f / (float)it;
                        const float w = (float)(it - 1);
                        acc = mk((acc.x * w + c.x) * inv_it, (acc.y * w + c.y) * inv_it,
                                 (acc.z * w + c.z) * inv_it);
                    } else {
                        acc = acc + c;
                    }
                    if (RECORD) {
                        if (p.pixel_rgb) {
                            float *pr = p.pixel_rgb + (size_t)rect_pix * 3;
                            pr[0] = c.x; pr[1] = c.y; pr[2] = c.z;
                        }
                    }
                    c = mk(0.f, 0.f, 0.f);
                    sub = 0;
                    if (++it == it_end) done = true;
                }
            }
        }
    }

    if (valid && p.buffer && p.n_iterations) {
        float *px = p.buffer + (size_t)pix * 3;
        px[0] = acc.x; px[1] = acc.y; px[2] = acc.z;
    }
    const uint32_t total = wave_sum(nrays);
    if (lane == 0 && p.ray_counter && total) atomicAdd(p.ray_counter, (unsigned long long)total);
}

// ---- arithmetic self-test kernels ---------------------------------------------------------------
__device__ __forceinline__ float selftest_op(int op, float a, float b)
{
    float s, c;
    switch (op) {
    case 0: return pt_sqrt(a);
    case 1: return pt_rcp(a);
    case 2: return pt_div(a, b);
    case 3: pt_sincosf(a, s, c); return s;
    default: pt_sincosf(a, s, c); return c;
    }
}

__global__ void selftest_eval_kernel(int op, const float *a, const float *b, float *out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = selftest_op(op, a[i], b ? b[i] : 0.f);
}

// Compares the short sequences with the compiler's generic IEEE expansions: every one of the
// 2^32 bit patterns for sqrt and 1/x, 2^32 Philox-random operand pairs for a/b.
__global__ void selftest_exhaustive_kernel(int op, unsigned long long *mismatches)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // grid covers 2^24 threads
    uint32_t bad = 0;
    for (uint32_t k = 0; k < 256; ++k) {
        const uint32_t bits = (k << 24) | i;
        float got, want;
        if (op == 0) { const float x = __uint_as_float(bits); got = pt_sqrt(x); want = __builtin_sqrtf(x); }
        else if (op == 1) { const float x = __uint_as_float(bits); got = pt_rcp(x); want = 1.f / x; }
        else {
            uint32_t r[4];
            philox4x32_10(bits, 0u, 0u, 0x5e1f7e57u, 1u, 2u, r);
            // exponents confined to the fast range half of the time, anything otherwise
            float a = __uint_as_float(r[0]), b = __uint_as_float(r[1]);
            if (r[2] & 1u) {
                a = __uint_as_float((r[0] & 0x807fffffu) | ((100u + (r[2] >> 8) % 54u) << 23));
                b = __uint_as_float((r[1] & 0x807fffffu) | ((100u + (r[3] >> 8) % 54u) << 23));
            }
            got = pt_div(a, b); want = a / b;
        }
        const bool both_nan = (got != got) && (want != want);
        if (!both_nan && __float_as_uint(got) != __float_as_uint(want)) ++bad;
    }
    const uint32_t total = wave_sum(bad);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(mismatches, (unsigned long long)total);
}

} // namespace

hipError_t launch_selftest_eval(int op, const float *a, const float *b, float *out, uint64_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(selftest_eval_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, op, a, b, out, n);
    return hipGetLastError();
}

hipError_t launch_selftest_exhaustive(int op, unsigned long long *mismatches, hipStream_t stream)
{
    hipLaunchKernelGGL(selftest_exhaustive_kernel, dim3(1u << 16), dim3(256), 0, stream, op, mismatches);
    return hipGetLastError();
}

size_t render_lds_bytes(const RenderParams &p, bool tiled)
{
    if (tiled) return (size_t)kTileSpheres * sizeof(float4);
    return (size_t)p.n_prims * 2 * sizeof(float4) + (size_t)p.n_mats * (sizeof(float4) + sizeof(int)) +
           (size_t)p.n_lights * sizeof(float4);
}

hipError_t launch_render(const RenderParams &p, uint32_t grid_x, uint32_t grid_y, bool tiled,
                         bool record, hipStream_t stream)
{
    const dim3 grid(grid_x, grid_y), block(kBlockThreads);
    const size_t lds = render_lds_bytes(p, tiled);
    if (tiled) {
        if (record) hipLaunchKernelGGL((render_kernel<true, true>), grid, block, lds, stream, p);
        else        hipLaunchKernelGGL((render_kernel<true, false>), grid, block, lds, stream, p);
    } else {
        if (record) hipLaunchKernelGGL((render_kernel<false, true>), grid, block, lds, stream, p);
        else        hipLaunchKernelGGL((render_kernel<false, false>), grid, block, lds, stream, p);
    }
    return hipGetLastError();
}

} // namespace ptmi
