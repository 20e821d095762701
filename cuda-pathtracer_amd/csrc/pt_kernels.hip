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
#include "pt_nee.hpp"

#include <mutex>

namespace ptmi {

namespace {

template <bool TILED, bool RECORD, bool RING, bool BVH>
__global__ __launch_bounds__(kBlockThreads) void render_kernel(const RenderParams p)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const SceneView sc = stage_scene<TILED>(p.scene, lds_raw, tid);

    // ---- pixel owned by this lane ----------------------------------------------------------
    // blockIdx.y enumerates the 16-row block rows of the bands this instance owns.
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

    // ---- camera-ray ring (small resident scenes) -----------------------------------------------
    // Generating a camera ray costs two thirds of a bounce of shading, and in every loop pass
    // only the ~quarter of the lanes whose path just ended need one: done inline, the whole
    // wave pays for it on every pass.  Instead each lane keeps a ring of 8 rays (two
    // pixel-iterations) in LDS.  Only when some lane has run dry does the wave enter the
    // generation block, and then EVERY lane with room generates its next pixel-iteration
    // (4 rays, 2 Philox blocks): the block runs every ~7 passes with about half the lanes active.
    // The ring costs 96 B of LDS per lane (6 instead of 8 workgroups per CU), which only pays
    // while ray generation is a large share of a pass, i.e. for scenes of a few spheres.
    float *ring = reinterpret_cast<float *>(lds_raw + scene_lds_bytes_dev(p.scene));
    uint32_t gen = 0;    // pixel-iterations generated into the ring so far
    uint32_t cons = 0;   // paths started so far; path q is (iteration q/4, sub-sample q%4)

    for (;;) {
        // ---- path regeneration ---------------------------------------------------------------
        // regen_min_idle = 1: any idle lane restarts at once; 64: lanes restart only when the
        // whole wave is idle (the classic one-path-per-lane-at-a-time megakernel).
        bool want = !done && !alive;
        if (p.regen_min_idle > 1) {
            const int idle = __popcll(__ballot(!alive));
            want = want && idle >= p.regen_min_idle;
        }
        if (RING) {
            if (__any(want && cons == gen * 4u)) {
                if (!done && gen < p.n_iterations && gen * 4u - cons <= 4u) {
                    PT_LS(0);
                    V3 dirs[4];
                    camera_rays4(p.cam, x, y, pix, p.first_iteration + gen, k0, k1, dirs);
                    const uint32_t base = (gen & 1u) * 4u;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        float *slot = ring + ((base + s) * 3u) * kBlockThreads + tid;
                        slot[0] = dirs[s].x; slot[kBlockThreads] = dirs[s].y; slot[2 * kBlockThreads] = dirs[s].z;
                    }
                    ++gen;
                }
            }
        }
        if (want) {
            PT_LS(1);
            o = cam_pos;
            if (RING) {
                const float *slot = ring + ((cons & 7u) * 3u) * kBlockThreads + tid;
                d = mk(slot[0], slot[kBlockThreads], slot[2 * kBlockThreads]);
                ++cons;
            } else {
                d = camera_ray(p.cam, x, y, pix, it, sub, k0, k1);
            }
            T = mk(1.f, 1.f, 1.f);
            L = mk(0.f, 0.f, 0.f);
            b = 0;
            alive = true;
        }
        if (TILED && !BVH) {
            if (!__syncthreads_or(alive)) break;
        } else {
            if (!__any(alive)) break;
        }

        float tmax;
        int best;
        intersect_scene<TILED, BVH>(p.scene, sc, lds_raw, tid, alive, o, d, tmax, best);

        PT_LS(4);
        if (alive) {
            PT_LS(17);
            ++nrays;
            if (RECORD) {
                if (p.hit_ids)
                    p.hit_ids[((size_t)rect_pix * 4 + sub) * (size_t)(mb + 1) + b] = best;
            }
            bool end = true;
            if (best >= 0) {
                const float4 s = TILED ? p.scene.sph[best] : sc.sph[best];
                end = shade_bounce(sc, s, best, tmax, pix, it, k0, k1, mb, o, d, T, L, b);
            }
            if (end) {
                if (RECORD) {
                    if (p.path_rad) {
                        float *pr = p.path_rad + ((size_t)rect_pix * 4 + sub) * 3;
                        pr[0] = L.x; pr[1] = L.y; pr[2] = L.z;
                    }
                }
                PT_LS(18);
                c = c + L;
                alive = false;
                if (++sub == 4) {
                    PT_LS(19);
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

// ---- scratch blocks of the work-stealing megakernel --------------------------------------------------
// Takes a free block of the pool (layout: pt_kernels.hpp) for the calling workgroup, from the
// sub-pool of the XCD it runs on: a block is then only ever read and written through that XCD's
// L2, so what a workgroup left in it needs no write-back before the next one may use it.
// Returns kNoStealBlock when the sub-pool has no free block (the workgroup then does not steal);
// bounded: at most four words are tried, each until it is full.
constexpr uint32_t kNoStealBlock = 0xffffffffu;

PT_DEV uint32_t steal_block_acquire(unsigned long long *mask, uint32_t hint)
{
    // XCC_ID (hardware register 20), bits 3:0: the XCD this wave runs on
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & (kStealPoolXcc - 1u);
    unsigned long long *m = mask + xcc * kStealMaskWordsPerXcc;
    const uint32_t rot = (hint >> 2) & 63u;   // workgroups that start together try different bits first
    for (uint32_t k = 0; k < kStealPoolPerXcc / 64u; ++k) {
        const uint32_t w = (hint + k) & (kStealPoolPerXcc / 64u - 1u);
        unsigned long long seen = __hip_atomic_load(m + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (~seen) {
            const unsigned long long avail = ~seen;
            const unsigned long long rotated = (avail >> rot) | (rot ? avail << (64u - rot) : 0ull);
            const uint32_t bit = ((uint32_t)__builtin_ctzll(rotated) + rot) & 63u;
            const unsigned long long b = 1ull << bit;
            const unsigned long long prev = __hip_atomic_fetch_or(m + w, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(prev & b)) return xcc * kStealPoolPerXcc + w * 64u + bit;
            seen = prev;
        }
    }
    return kNoStealBlock;
}

PT_DEV void steal_block_release(unsigned long long *mask, uint32_t block)
{
    const uint32_t xcc = block / kStealPoolPerXcc, in = block % kStealPoolPerXcc;
    __hip_atomic_fetch_and(mask + xcc * kStealMaskWordsPerXcc + (in >> 6), ~(1ull << (in & 63u)), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// parity hook: the colour pathtracer.cu:94-101 computes for pixel (x, y) of the traced rectangle
PT_DEV void record_pixel(const RenderParams &p, uint32_t x, uint32_t y, V3 cpix)
{
    if (p.pixel_rgb) {
        float *pr = p.pixel_rgb + ((size_t)(y - p.y0) * (p.x1 - p.x0) + (x - p.x0)) * 3;
        pr[0] = cpix.x; pr[1] = cpix.y; pr[2] = cpix.z;
    }
}

// ---- megakernel with work stealing inside the wave ---------------------------------------------------
// render_kernel gives every lane one pixel for the whole launch.  The pixels of a wave need
// different numbers of bounces, so lanes finish at different times and wait for the slowest one:
// at 1080p with 64 iterations 11 % of the lanes have no ray in the average loop pass (measured with
// the divergence counters, pt_lane_stats).  Here a lane without work of its own takes over a whole
// pixel-iteration (the four paths of one pixel in one iteration) from the lane with the most
// iterations left:
//   - always the LAST iteration the victim has not started (and has no camera rays for yet), so
//     the victim's own iterations stay a prefix 1..E of the range and are folded into the running
//     mean as they complete, exactly as before;
//   - the thief traces the four paths in order, which gives the same c = ((0+L0)+L1)+L2)+L3, and
//     leaves it in a per-pixel scratch slot in device memory (the stolen iterations are rare: a
//     few per lane and launch);
//   - when the wave has no work left, every lane folds the stolen iterations E..end of ITS pixel
//     from the scratch slots in ascending order: the accumulation order of the reference
//     (pathtracer.cu:104-109) is kept and the image is bit-identical.
// At most kStealSlots iterations are taken from one pixel per launch.  Stealing costs nothing
// until the first lane of a wave runs out of work (one wave-wide test per pass).
// Waves per SIMD the register allocation aims at: 5 for the NEE integrator (96 VGPRs), 6 where LDS
// (camera-ray ring, sphere tile) allows no more anyway, 7 (72 VGPRs, no spills; 8 would spill) for the plain resident scenes: +3 % on the
// 71-sphere scene.
//
// RECORD: the parity hook's outputs (hit indices, per-path radiance, pixel colours) are written as
// well, so that pt_debug_trace reports what THIS kernel computes, stolen paths included.
template <bool RING, bool TILED, bool NEE, bool BVH, bool UNIT1, bool RECORD>
__global__ __launch_bounds__(kBlockThreads, BVH ? 4 : (NEE ? 5 : ((RING || TILED) ? 6 : 7))) void render_steal_kernel(const RenderParams p)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ uint32_t s_wg[2];   // start-order ticket, scratch block of this workgroup

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        s_wg[0] = p.n_chunks > 1 ? atomicAdd(p.tickets, 1u) : 0u;
        s_wg[1] = p.steal_scratch ? steal_block_acquire(p.steal_mask, blockIdx.y * gridDim.x + blockIdx.x) : kNoStealBlock;
    }
    __syncthreads();
    const uint32_t steal_block = s_wg[1];

    // ---- tile and iteration range of this workgroup ------------------------------------------------------
    // Plain launch: the tile is the block index, the workgroup renders all iterations.
    // Chunked launch (p.n_chunks > 1, grid.y = n_chunks x tile rows): the iterations are cut into
    // n_chunks consecutive ranges rendered by different workgroups one after the other, each
    // continuing from the frame-buffer values its predecessor left (what consecutive launches would
    // do).  More, shorter workgroups leave fewer SIMDs idle while the last ones of a launch finish.
    // Workgroups take a ticket in the order they START: tickets 0 .. tiles-1 are range 0 of every
    // tile, and so on, so the workgroup a range waits for has always started already and never
    // waits for a later ticket itself -- no deadlock, whatever order blocks are dispatched in.
    uint32_t bx = blockIdx.x, by = blockIdx.y, chunk = 0, tile = 0;
    uint32_t first_it = p.first_iteration, n_it = p.n_iterations;
    if (p.n_chunks > 1) {
        const uint32_t ticket = s_wg[0];
        const uint32_t n_tiles = gridDim.x * (gridDim.y / p.n_chunks);
        chunk = ticket / n_tiles;
        tile = ticket - chunk * n_tiles;
        by = tile / gridDim.x;
        bx = tile - by * gridDim.x;
        const uint32_t per = (p.n_iterations + p.n_chunks - 1) / p.n_chunks;   // host: no range is empty
        first_it = p.first_iteration + chunk * per;
        n_it = p.n_iterations - chunk * per < per ? p.n_iterations - chunk * per : per;
        if (chunk > 0) {
            if (tid == 0) {
                while (__hip_atomic_load(p.tickets + 1 + tile, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < chunk)
                    __builtin_amdgcn_s_sleep(16);
            }
            __syncthreads();
        }
    }
    const SceneView sc = stage_scene<TILED>(p.scene, lds_raw, tid);

    const uint32_t blocks_per_band = p.band_rows >> 4;
    const uint32_t band_local = by / blocks_per_band;
    const uint32_t band = band_local * p.band_world + p.band_rank;
    const uint32_t yb = p.y0 + band * p.band_rows + (by - band_local * blocks_per_band) * 16u;
    const uint32_t wave_x0 = p.x0 + bx * 16u + (uint32_t)((wave & 1) * 8);
    const uint32_t wave_y0 = yb + (uint32_t)((wave >> 1) * 8);
    const uint32_t x = wave_x0 + (uint32_t)(lane & 7), y = wave_y0 + (uint32_t)(lane >> 3);
    const bool valid = x < p.x1 && y < p.y1;
    const uint32_t pix = y * p.cam.width + x;

    const V3 cam_pos = mk(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
    const uint32_t k0 = (uint32_t)p.seed, k1 = (uint32_t)(p.seed >> 32);
    const int mb = p.max_bounces;
    const uint32_t first = first_it;
    // ---- this lane's pixel -------------------------------------------------------------------------
    // The paths of a pixel are numbered g = (iteration - first) * 4 + sub-sample, n_paths = 4 * n_it in
    // all, and traced in that order.  Work changes hands in UNITS of `unit` consecutive paths: 4 (a
    // whole pixel-iteration, the thief hands back the 4-path sum c) for long launches, 1 (a single
    // path, the thief hands back its radiance L) for launches of a few iterations, where a lane
    // has too few whole iterations to give away.  A lane's own paths are [0, own_end); thieves
    // take units off its end.
    const uint32_t n_paths = n_it * 4u;
    constexpr uint32_t unit = UNIT1 ? 1u : 4u;
    uint32_t next_own = 0;                    // next own path to start
    uint32_t own_end = valid ? n_paths : 0u;
    V3 acc = mk(0.f, 0.f, 0.f);
    if (valid && (!p.overwrite || chunk > 0) && p.buffer) {
        const float *px = p.buffer + (size_t)pix * 3;
        acc = mk(px[0], px[1], px[2]);
    }
    // ---- the path in hand (own or stolen) ----------------------------------------------------------
    bool has_task = false, alive = false;
    uint32_t tpix = pix, tg = 0, tit = first;   // pixel, path number and iteration of the path in hand
    // -1: the lane's own pixel; else the lane (of this wave) the unit was taken from, + 64 when its
    // camera rays sit in that lane's ring
    int t_from = -1;
    // Running 4-path sum of the OWN pixel's current iteration (pathtracer.cu:95-101); a thief of whole
    // iterations has none of its own left and builds the stolen iteration's sum here, a thief of
    // single paths leaves it alone (it may hold the first paths of the lane's last iteration).
    V3 c = mk(0.f, 0.f, 0.f);
    V3 o = cam_pos, d = mk(0.f, 0.f, 1.f), T = mk(1.f, 1.f, 1.f), L = mk(0.f, 0.f, 0.f);
    int b = 0;
    uint32_t nrays = 0;
    bool stealing = steal_block != kNoStealBlock;   // wave-uniform; cleared once nothing is left to steal
    // this wave's part of the workgroup's scratch block: [slot][lane of the wave][3]
    float *const steal_wave = p.steal_scratch + (size_t)(stealing ? steal_block : 0u) * kStealBlockFloats + (size_t)(wave * 64) * 3;
    // NEE + MIS integrator (see render_nee_kernel): a diffuse vertex parks its direct-light term and
    // sends the shadow ray through the next pass's Intersect
    NeeState st;
    st.shadow = false; st.end_after = false; st.specular_bounce = false; st.bsdf_pdf_prev = 0.f;
    st.sh_d = mk(0.f, 0.f, 1.f); st.sh_tmax = PT_INF; st.sh_prim = -1; st.pending = mk(0.f, 0.f, 0.f);

    // camera-ray ring of the lane's OWN iterations, see render_kernel
    float *ring = reinterpret_cast<float *>(lds_raw + scene_lds_bytes_dev(p.scene));
    uint32_t gen = 0;    // own pixel-iterations generated into the ring so far
    uint32_t cons = 0;   // own paths started so far

    if (next_own < own_end) { has_task = true; ++next_own; }   // the first own path (tpix, tg, tit, t_from are set for it)

    for (;;) {
        // ---- a lane that has finished its own paths takes the last unstarted unit of the lane with
        //      the most left ----------------------------------------------------------------------------------
        if (stealing && __any(!has_task)) {
            PT_LS(20);
            for (;;) {
                // units this lane could give away: whole ones it has not started
                const uint32_t first_free = (next_own + unit - 1u) & ~(unit - 1u);
                const uint32_t spare = (own_end > first_free && (n_paths - own_end) / unit < kStealSlots)
                                           ? (own_end - first_free) / unit : 0u;
                uint32_t key = (spare << 6) | (uint32_t)lane;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const uint32_t other = (uint32_t)__shfl_xor((int)key, off, 64);
                    key = other > key ? other : key;
                }
                key = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
                if ((key >> 6) == 0u) { stealing = false; break; }
                const unsigned long long idle = __ballot(!has_task);
                if (!idle) break;
                const int thief = __ffsll(idle) - 1;
                const int victim = (int)(key & 63u);
                const uint32_t v_g = (uint32_t)__builtin_amdgcn_readlane((int)own_end, victim) - unit;
                const uint32_t v_pix = (uint32_t)__builtin_amdgcn_readlane((int)pix, victim);
                // the victim's ring may hold the camera rays of the unit already; it will not write to
                // those slots again (it generates no iteration that starts at or beyond own_end)
                const uint32_t v_gen = (uint32_t)__builtin_amdgcn_readlane((int)gen, victim);
                if (lane == victim) own_end = v_g;
                if (lane == thief) {
                    PT_LS(21);
                    has_task = true;
                    tpix = v_pix; tg = v_g; tit = first + (v_g >> 2);
                    t_from = victim + ((RING && (v_g >> 2) < v_gen) ? 64 : 0);
                    if (unit == 4u) c = mk(0.f, 0.f, 0.f);
                }
            }
        }

        // ---- start the path in hand --------------------------------------------------------------------------
        const bool want = has_task && !alive;
        if (RING) {
            if (__any(want && t_from < 0 && cons == gen * 4u)) {
                if (gen * 4u < own_end && gen * 4u - cons <= 4u) {
                    V3 dirs[4];
                    camera_rays4(p.cam, x, y, pix, first + gen, k0, k1, dirs);
                    const uint32_t base = (gen & 1u) * 4u;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        float *slot = ring + ((base + s) * 3u) * kBlockThreads + tid;
                        slot[0] = dirs[s].x; slot[kBlockThreads] = dirs[s].y; slot[2 * kBlockThreads] = dirs[s].z;
                    }
                    ++gen;
                }
            }
        }
        if (want) {
            o = cam_pos;
            if (RING && t_from < 0) {
                const float *slot = ring + ((cons & 7u) * 3u) * kBlockThreads + tid;
                d = mk(slot[0], slot[kBlockThreads], slot[2 * kBlockThreads]);
                ++cons;
            } else if (RING && t_from >= 64) {
                const float *slot = ring + ((tg & 7u) * 3u) * kBlockThreads + (tid - lane + (t_from - 64));
                d = mk(slot[0], slot[kBlockThreads], slot[2 * kBlockThreads]);
            } else {
                const int from = t_from < 0 ? lane : t_from;
                d = camera_ray(p.cam, wave_x0 + (uint32_t)(from & 7), wave_y0 + (uint32_t)(from >> 3), tpix, tit,
                               (int)(tg & 3u), k0, k1);
            }
            T = mk(1.f, 1.f, 1.f);
            L = mk(0.f, 0.f, 0.f);
            b = 0;
            alive = true;
            if (NEE) { st.shadow = false; st.specular_bounce = false; }
        }
        // streaming scenes: the scan stages sphere tiles with workgroup barriers, so the four waves
        // leave the loop together
        if (TILED && !BVH) {
            if (!__syncthreads_or(alive)) break;
        } else {
            if (!__any(alive)) break;
        }

        float tmax;
        int best;
        const bool shadow = NEE && alive && st.shadow;
        if (NEE) intersect_scene<TILED, BVH>(p.scene, sc, lds_raw, tid, alive, o, shadow ? st.sh_d : d, tmax, best,
                                            shadow ? st.sh_tmax : PT_INF);
        else     intersect_scene<TILED, BVH>(p.scene, sc, lds_raw, tid, alive, o, d, tmax, best);

        PT_LS(4);
        if (alive) {
            PT_LS(17);
            ++nrays;
            // parity hook: index of the path in hand within the traced rectangle
            size_t rec_path = 0;
            if (RECORD) {
                const uint32_t from = t_from < 0 ? (uint32_t)lane : (uint32_t)(t_from & 63);
                const uint32_t tx = wave_x0 + (from & 7u), ty = wave_y0 + (from >> 3);
                rec_path = ((size_t)(ty - p.y0) * (p.x1 - p.x0) + (tx - p.x0)) * 4 + (tg & 3u);
                if (p.hit_ids && !shadow) p.hit_ids[rec_path * (size_t)(mb + 1) + b] = best;
            }
            bool end = true;
            if (shadow) {
                const bool visible = st.sh_prim >= 0 ? best == st.sh_prim : best < 0;
                if (visible) L = L + st.pending;
                st.shadow = false;
                end = st.end_after;
            } else if (best >= 0) {
                const float4 s = TILED ? p.scene.sph[best] : sc.sph[best];
                if (NEE) {
                    end = shade_bounce_nee(p.scene, sc, s, best, tmax, tpix, tit, k0, k1, mb, o, d, T, L, b, st);
                    if (st.shadow) { st.end_after = end; end = false; }
                } else {
                    end = shade_bounce(sc, s, best, tmax, tpix, tit, k0, k1, mb, o, d, T, L, b);
                }
            }
            if (end) {
                alive = false;
                const bool last_of_iteration = (tg & 3u) == 3u;
                if (RECORD) {
                    if (p.path_rad) {
                        float *pr = p.path_rad + rec_path * 3;
                        pr[0] = L.x; pr[1] = L.y; pr[2] = L.z;
                    }
                }
                if (t_from < 0) {
                    // own path: the running sum, and the running mean once the iteration is complete
                    c = c + L;
                    if (last_of_iteration) {
                        V3 cpix;
                        acc = accumulate_pixel(acc, c, tit, p.accum_mode, &cpix);
                        if (RECORD) record_pixel(p, x, y, cpix);
                        c = mk(0.f, 0.f, 0.f);
                    }
                    has_task = next_own < own_end;
                    tg = next_own;
                    next_own += has_task ? 1u : 0u;
                    tit = first + (tg >> 2);
                } else if (unit == 4u && !last_of_iteration) {
                    // stolen iteration: on to its next path
                    c = c + L;
                    ++tg;
                } else {
                    // stolen unit complete: leave the 4-path sum, or the path's radiance, for the owner
                    const V3 v = unit == 4u ? c + L : L;
                    float *slot = steal_wave + ((size_t)((n_paths - tg - 1u) / unit) * kBlockThreads + (uint32_t)(t_from & 63)) * 3;
                    slot[0] = v.x; slot[1] = v.y; slot[2] = v.z;
                    if (unit == 4u) c = mk(0.f, 0.f, 0.f);
                    has_task = false;   // a thief has no paths of its own left
                    t_from = -1;
                    tpix = pix;
                }
            }
        }
    }

    // ---- the units other lanes of this wave rendered for this pixel, in order -----------------------------------
    if (own_end < n_paths && valid) {
        __threadfence_block();   // their results were stored by this same wave
        for (uint32_t g = own_end; g < n_paths; g += unit) {
            const float *slot = steal_wave + ((size_t)((n_paths - g - 1u) / unit) * kBlockThreads + (uint32_t)lane) * 3;
            // read past the vector cache: the line may have been cached before a thief wrote to it
            const V3 v = mk(__hip_atomic_load(slot + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                            __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                            __hip_atomic_load(slot + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            V3 cpix;
            if (unit == 4u) {
                acc = accumulate_pixel(acc, v, first + (g >> 2), p.accum_mode, &cpix);
                if (RECORD) record_pixel(p, x, y, cpix);
            } else {
                c = c + v;
                if ((g & 3u) == 3u) {
                    acc = accumulate_pixel(acc, c, first + (g >> 2), p.accum_mode, &cpix);
                    if (RECORD) record_pixel(p, x, y, cpix);
                    c = mk(0.f, 0.f, 0.f);
                }
            }
        }
    }
    if (valid && p.buffer && n_it) {
        float *px = p.buffer + (size_t)pix * 3;
        px[0] = acc.x; px[1] = acc.y; px[2] = acc.z;
    }
    if (p.n_chunks > 1) {
        // hand the tile's frame-buffer values to the workgroup that renders its next range
        __threadfence();
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(p.tickets + 1 + tile, chunk + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (steal_block != kNoStealBlock) {
        // every wave is through with the block (the barrier also waits for their stores to it)
        __syncthreads();
        if (tid == 0) steal_block_release(p.steal_mask, steal_block);
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

// sample_material row by row: in [n][8] = wo.xyz, n.xyz, su, sv; out [n][8] = f.rgb, wi.xyz, pdf, 0
__global__ void selftest_material_kernel(int type, float4 color_coef, float inv_coef, const float *in, float *out, uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *x = in + 8 * i;
    float *y = out + 8 * i;
    V3 wi = mk(0.f, 0.f, 0.f);
    float pdf = 0.f;
    const V3 f = sample_material(type, mk(color_coef.x, color_coef.y, color_coef.z), color_coef.w, inv_coef, mk(x[0], x[1], x[2]),
                                 wi, pdf, mk(x[3], x[4], x[5]), x[6], x[7]);
    y[0] = f.x; y[1] = f.y; y[2] = f.z; y[3] = wi.x; y[4] = wi.y; y[5] = wi.z; y[6] = pdf; y[7] = 0.f;
}

// Compares the short sequences with the compiler's generic IEEE expansions: every one of the
// 2^32 bit patterns for sqrt (op 0; op 3: the sphere test's variant) and 1/x, 2^32 Philox-random
// operand pairs for a/b.
__global__ void selftest_exhaustive_kernel(int op, unsigned long long *mismatches)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // grid covers 2^24 threads
    uint32_t bad = 0;
    for (uint32_t k = 0; k < 256; ++k) {
        const uint32_t bits = (k << 24) | i;
        float got, want;
        if (op == 0) { const float x = __uint_as_float(bits); got = pt_sqrt(x); want = __builtin_sqrtf(x); }
        else if (op == 1) { const float x = __uint_as_float(bits); got = pt_rcp(x); want = 1.f / x; }
        else if (op == 3) {
            // the sphere test's variant: everything but +infinity (there: NaN or +infinity, no finite value)
            const float x = __uint_as_float(bits);
            got = pt_sqrt_discriminant(x); want = __builtin_sqrtf(x);
            if (bits == 0x7f800000u) got = (got != got || got == want) ? want : 0.f;
        }
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

hipError_t launch_selftest_material(int type, float4 color_coef, float inv_coef, const float *in, float *out, uint64_t n, hipStream_t stream)
{
    hipLaunchKernelGGL(selftest_material_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, type, color_coef, inv_coef, in, out, n);
    return hipGetLastError();
}

hipError_t launch_selftest_exhaustive(int op, unsigned long long *mismatches, hipStream_t stream)
{
    hipLaunchKernelGGL(selftest_exhaustive_kernel, dim3(1u << 16), dim3(256), 0, stream, op, mismatches);
    return hipGetLastError();
}

// Lane statistics of the megakernels in this file (measurement builds only): reads and clears the
// 32 (executions, active lanes) pairs; all zero in the product build.
hipError_t read_lane_stats(unsigned long long out[64])
{
#ifdef PT_LANE_STATS
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lane_stats), 64 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    static const unsigned long long zeros[64] = {};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stats), zeros, sizeof zeros);
#else
    for (int i = 0; i < 64; ++i) out[i] = 0;
    return hipSuccess;
#endif
}

// ---- statically allocated scratch of the entry point that allocates nothing ------------------------------
// Scratch of the work-stealing megakernel for pt_run_n_noalloc: the pool of scratch blocks, its
// occupancy bits, and the tickets of chunked launches (start-order counter + one word per
// 16x16-pixel tile, so frames of up to kStaticTicketWords - 1 tiles), 48.3 MiB in all.  Every GPU
// the library runs on holds its own copy at its own address until the process ends.  In host code
// the array's name is only a handle for hipGetSymbolAddress; kernels get the looked-up address as
// an argument.
static __device__ __attribute__((used)) __align__(256) unsigned char scratch[kStaticScratchBytes];

hipError_t static_steal_scratch(int device, StealScratch *out)
{
    constexpr int kMaxDevices = 64;
    static std::mutex mutex;
    static void *address[kMaxDevices];   // per device: that device's copy of `scratch`
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(mutex);
    if (!address[device]) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) return e;
        if (current != device) return hipErrorInvalidDevice;   // the symbol resolves on the current device
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        address[device] = ptr;
    }
    unsigned char *base = static_cast<unsigned char *>(address[device]);
    out->pool = reinterpret_cast<float *>(base);
    out->mask = reinterpret_cast<unsigned long long *>(base + kStealPoolBytes);
    out->tickets = reinterpret_cast<uint32_t *>(base + kStealPoolBytes + kStealMaskBytes);
    out->ticket_words = kStaticTicketWords;
    return hipSuccess;
}

size_t scene_lds_bytes(const SceneParams &s, bool tiled)
{
    if (tiled) return (size_t)kTileSpheres * sizeof(float4);
    const size_t n = (size_t)s.n_prims * 2 * sizeof(float4) + (size_t)s.n_mats * (sizeof(float4) + sizeof(int2)) +
                     (size_t)s.n_lights * sizeof(float4) + (size_t)s.n_tris * 4 * sizeof(float4);
    return (n + 15) & ~(size_t)15;
}

// LDS of the megakernel: the scene, plus (resident scenes only) the camera-ray ring,
// 8 rays x 3 floats per lane.
static size_t render_lds_bytes(const SceneParams &s, bool tiled, bool ring)
{
    return scene_lds_bytes(s, tiled) + (ring ? (size_t)8 * 3 * kBlockThreads * sizeof(float) : 0);
}

// The work-stealing megakernel (linear-scan scenes, p.steal_scratch set); nee: the NEE + MIS integrator.
hipError_t launch_render_steal(const RenderParams &p, uint32_t grid_x, uint32_t grid_y, bool tiled, bool nee,
                               bool record, hipStream_t stream)
{
    // BVH scenes keep the scene in device memory (TILED staging without the sphere tile) and the
    // per-lane traversal stacks in LDS.
    const bool bvh = p.scene.bvh.n_nodes > 0;
    // no camera-ray ring under NEE: its state needs 96 VGPRs (5 waves per SIMD), the ring's index
    // arithmetic 12 more (4 waves), which costs more than the ring saves (measured: 85.6 vs 82.4 ms)
    const bool ring = !nee && !bvh && !tiled && p.scene.n_prims <= kMaxRingSpheres && (p.small_scene_opts & 1);
    const size_t lds = bvh ? bvh_stack_lds_bytes(p.scene.bvh) : render_lds_bytes(p.scene, tiled, ring);
    const dim3 grid(grid_x, grid_y * (p.n_chunks > 1 ? p.n_chunks : 1u)), block(kBlockThreads);
#define PT_LAUNCH_STEAL(G, T, N, B)                                                                              \
    do {                                                                                                          \
        if (record) {                                                                                             \
            if (p.steal_unit == 1) hipLaunchKernelGGL((render_steal_kernel<G, T, N, B, true, true>), grid, block, lds, stream, p); \
            else hipLaunchKernelGGL((render_steal_kernel<G, T, N, B, false, true>), grid, block, lds, stream, p);  \
        } else {                                                                                                  \
            if (p.steal_unit == 1) hipLaunchKernelGGL((render_steal_kernel<G, T, N, B, true, false>), grid, block, lds, stream, p); \
            else hipLaunchKernelGGL((render_steal_kernel<G, T, N, B, false, false>), grid, block, lds, stream, p); \
        }                                                                                                         \
    } while (0)
    if (bvh) {
        if (nee) PT_LAUNCH_STEAL(false, true, true, true); else PT_LAUNCH_STEAL(false, true, false, true);
    } else if (nee) {
        if (tiled) PT_LAUNCH_STEAL(false, true, true, false); else PT_LAUNCH_STEAL(false, false, true, false);
    } else {
        if (tiled) PT_LAUNCH_STEAL(false, true, false, false); else if (ring) PT_LAUNCH_STEAL(true, false, false, false); else PT_LAUNCH_STEAL(false, false, false, false);
    }
#undef PT_LAUNCH_STEAL
    return hipGetLastError();
}

hipError_t launch_render(const RenderParams &p, uint32_t grid_x, uint32_t grid_y, bool tiled,
                         bool record, hipStream_t stream)
{
    const dim3 grid(grid_x, grid_y), block(kBlockThreads);
    // Small resident scenes keep a camera-ray ring (p.small_scene_opts bit 0; experiments can
    // switch it off).  A branch-free variant of the scan was measured 3.5 % slower and dropped.
    const bool ring = !tiled && p.scene.n_prims <= kMaxRingSpheres && (p.small_scene_opts & 1);
    const bool bvh = p.scene.bvh.n_nodes > 0;   // BVH kernels read the scene from global memory
    // Work stealing inside the wave: linear-scan scenes (staged in LDS or streamed), plain renders, regeneration on
    if (p.steal_scratch && p.regen_min_idle <= 1) return launch_render_steal(p, grid_x, grid_y, tiled, false, record, stream);
    const size_t lds = bvh ? bvh_stack_lds_bytes(p.scene.bvh) : render_lds_bytes(p.scene, tiled, ring);
#define PT_LAUNCH(T, R, G, B) hipLaunchKernelGGL((render_kernel<T, R, G, B>), grid, block, lds, stream, p)
    if (bvh) {
        if (record) PT_LAUNCH(true, true, false, true); else PT_LAUNCH(true, false, false, true);
    } else if (tiled) {
        if (record) PT_LAUNCH(true, true, false, false); else PT_LAUNCH(true, false, false, false);
    } else if (ring) {
        if (record) PT_LAUNCH(false, true, true, false); else PT_LAUNCH(false, false, true, false);
    } else {
        if (record) PT_LAUNCH(false, true, false, false); else PT_LAUNCH(false, false, false, false);
    }
#undef PT_LAUNCH
    return hipGetLastError();
}

} // namespace ptmi
