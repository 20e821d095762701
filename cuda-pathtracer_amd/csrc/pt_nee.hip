// Render kernel of the opt-in NEE + MIS integrator (PT_INTEGRATOR_NEE_MIS).
//
// Same mapping and accumulation as render_kernel (pt_kernels.hip): a lane owns one pixel for the
// whole launch, walks its paths in the reference's order and keeps the accumulator in registers;
// idle lanes start their next path at the top of the loop body.  The body is still ONE Intersect:
// a diffuse vertex that found a usable light sample parks its direct-light contribution, sends
// the shadow ray through the next pass's Intersect, and only then continues (or ends) its path,
// so shadow rays and path rays of different lanes share the same scan / BVH traversal.
#include "pt_nee.hpp"

namespace ptmi {

namespace {

template <bool TILED, bool RECORD, bool BVH>
__global__ __launch_bounds__(kBlockThreads) void render_nee_kernel(const RenderParams p)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const SceneView sc = stage_scene<TILED>(p.scene, lds_raw, tid);

    const uint32_t blocks_per_band = p.band_rows >> 4;
    const uint32_t band_local = blockIdx.y / blocks_per_band;
    const uint32_t band = band_local * p.band_world + p.band_rank;
    const uint32_t yb = p.y0 + band * p.band_rows + (blockIdx.y - band_local * blocks_per_band) * 16u;
    const uint32_t x = p.x0 + blockIdx.x * 16u + (uint32_t)((wave & 1) * 8 + (lane & 7));
    const uint32_t y = yb + (uint32_t)((wave >> 1) * 8 + (lane >> 3));
    const bool valid = x < p.x1 && y < p.y1;
    const uint32_t pix = y * p.cam.width + x;
    const uint32_t rect_pix = (y - p.y0) * (p.x1 - p.x0) + (x - p.x0);

    const V3 cam_pos = mk(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
    const uint32_t k0 = (uint32_t)p.seed, k1 = (uint32_t)(p.seed >> 32);
    const int mb = p.max_bounces;

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
    NeeState st;
    st.shadow = false; st.end_after = false; st.specular_bounce = false; st.bsdf_pdf_prev = 0.f;
    st.sh_d = mk(0.f, 0.f, 1.f); st.sh_tmax = PT_INF; st.sh_prim = -1; st.pending = mk(0.f, 0.f, 0.f);

    for (;;) {
        if (!done && !alive) {
            o = cam_pos;
            d = camera_ray(p.cam, x, y, pix, it, sub, k0, k1);
            T = mk(1.f, 1.f, 1.f);
            L = mk(0.f, 0.f, 0.f);
            b = 0;
            st.shadow = false;
            st.specular_bounce = false;
            alive = true;
        }
        if (TILED && !BVH) {
            if (!__syncthreads_or(alive)) break;
        } else {
            if (!__any(alive)) break;
        }

        float tmax;
        int best;
        const bool shadow = alive && st.shadow;
        intersect_scene<TILED, BVH>(p.scene, sc, lds_raw, tid, alive, o, shadow ? st.sh_d : d, tmax, best,
                                    shadow ? st.sh_tmax : PT_INF);

        if (alive) {
            ++nrays;
            bool end;
            if (shadow) {
                const bool visible = st.sh_prim >= 0 ? best == st.sh_prim : best < 0;
                if (visible) L = L + st.pending;
                st.shadow = false;
                end = st.end_after;
            } else {
                if (RECORD) {
                    if (p.hit_ids)
                        p.hit_ids[((size_t)rect_pix * 4 + sub) * (size_t)(mb + 1) + b] = best;
                }
                end = true;
                if (best >= 0) {
                    const float4 s = TILED ? p.scene.sph[best] : sc.sph[best];
                    end = shade_bounce_nee(p.scene, sc, s, best, tmax, pix, it, k0, k1, mb, o, d, T, L, b, st);
                    if (st.shadow) { st.end_after = end; end = false; }
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
                    V3 cpix;
                    acc = accumulate_pixel(acc, c, it, p.accum_mode, &cpix);
                    if (RECORD) {
                        if (p.pixel_rgb) {
                            float *pr = p.pixel_rgb + (size_t)rect_pix * 3;
                            pr[0] = cpix.x; pr[1] = cpix.y; pr[2] = cpix.z;
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

} // namespace

// Divergence counters of the kernel in this file (measurement builds only; every translation unit
// has its own copy of the counter array), added to out[64] and cleared.
hipError_t add_lane_stats_nee(unsigned long long out[64])
{
#ifdef PT_LANE_STATS
    unsigned long long v[64];
    hipError_t e = hipMemcpyFromSymbol(v, HIP_SYMBOL(g_lane_stats), sizeof v);
    if (e != hipSuccess) return e;
    for (int i = 0; i < 64; ++i) out[i] += v[i];
    static const unsigned long long zeros[64] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stats), zeros, sizeof zeros);
#else
    (void)out;
    return hipSuccess;
#endif
}

hipError_t launch_render_nee(const RenderParams &p, uint32_t grid_x, uint32_t grid_y, bool tiled, bool record,
                             hipStream_t stream)
{
    const dim3 grid(grid_x, grid_y), block(kBlockThreads);
    const bool bvh = p.scene.bvh.n_nodes > 0;   // BVH kernels read the scene from global memory
    // plain renders: the work-stealing megakernel (pt_kernels.hip)
    if (p.steal_scratch && p.regen_min_idle <= 1) return launch_render_steal(p, grid_x, grid_y, tiled, true, record, stream);
    const size_t lds = bvh ? bvh_stack_lds_bytes(p.scene.bvh) : scene_lds_bytes(p.scene, tiled);
#define PT_LAUNCH(T, R, B) hipLaunchKernelGGL((render_nee_kernel<T, R, B>), grid, block, lds, stream, p)
    if (bvh) {
        if (record) PT_LAUNCH(true, true, true); else PT_LAUNCH(true, false, true);
    } else if (tiled) {
        if (record) PT_LAUNCH(true, true, false); else PT_LAUNCH(true, false, false);
    } else {
        if (record) PT_LAUNCH(false, true, false); else PT_LAUNCH(false, false, false);
    }
#undef PT_LAUNCH
    return hipGetLastError();
}

} // namespace ptmi
