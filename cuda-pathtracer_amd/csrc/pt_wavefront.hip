// Wavefront (path-compaction) variant of the trace loop for gfx950.
//
// The megakernel (pt_kernels.hip) keeps a path in one lane from camera ray to termination.
// Here every bounce is its own launch over a compacted queue of live paths in device memory:
// each launch reads a path's state, runs one Intersect + one bounce of shading (the same device
// functions as the megakernel, so results are bit-identical), adds what the bounce collected to
// the path's radiance and appends the path to the next queue if it goes on.  All lanes of a wave start every
// launch alive and at the same bounce; the price is ~88 B of queue traffic per path-bounce
// and one launch per bounce.  pt_set_kernel_variant(h, PT_VARIANT_WAVEFRONT) selects it; which
// variant is faster depends on the scene (see the profiles).
//
// Queue layout: 64 sub-queues, each with its own counter on its own 128-byte line, so that the
// compaction atomics (one per wave) do not serialise on a single address.  Survivors of
// sub-queue q stay in sub-queue q, hence no sub-queue can outgrow its capacity.
#include "pt_nee.hpp"

#include <atomic>

namespace ptmi {

namespace {

// Waves per SIMD the registers of the direct-light kernels are allocated for (measured choices).
#ifndef PT_NEE_BVH_WAVES
#ifndef PT_BVH_WAVES
#define PT_BVH_WAVES 6   // waves per SIMD of the path integrator's wf_step_bvh_kernel
#endif
#define PT_NEE_BVH_WAVES 5   // waves per SIMD of the direct-light variant of wf_step_bvh_kernel
#endif
#ifndef PT_NEE_SHADOW_PASS_WAVES
#define PT_NEE_SHADOW_PASS_WAVES 6
#endif
#ifndef PT_NEE_POOL_WAVES
#define PT_NEE_POOL_WAVES 6   // waves per SIMD the direct-light pool kernel's registers are allocated for
#endif

// counts[bounce][q * kCountStride]: number of paths alive in sub-queue q entering `bounce`
PT_DEV uint32_t *count_ptr(uint32_t *counts, int bounce, uint32_t q)
{
    return counts + ((size_t)bounce * kWfSubQueues + q) * kWfCountStride;
}

// a / b through the float reciprocal of b: the estimate is off by at most one while the quotient
// stays below 2^20 and b below 2^30 (row, band and iteration numbers; the compiler's exact
// division is some 25 instructions, 64-bit operands more), `rem` gets a - q * b.
PT_DEV uint32_t div_small(uint32_t a, uint32_t b, uint32_t &rem)
{
    uint32_t q = (uint32_t)((float)a * __builtin_amdgcn_rcpf((float)b));
    int32_t r = (int32_t)(a - q * b);
    if (r < 0) { --q; r += (int32_t)b; }
    else if (r >= (int32_t)b) { ++q; r -= (int32_t)b; }
    rem = (uint32_t)r;
    return q;
}

// The batch covers the pixel rectangle [x0, x0 + rect_w) x [y0, y0 + rect_h) (the whole frame for a
// render, a part of it for the parity hook).  Paths are numbered over the rectangle's pixels in
// row-major order; a path carries its frame-wide pixel index (the Philox counter).
// With row bands (band_world > 1; full-width rectangle, y0 == 0) the rectangle's rows are the rows
// this instance owns, band after band: bands b with b % band_world == band_rank.
PT_DEV uint32_t rect_to_frame(const WavefrontParams &p, uint32_t local, uint32_t &x, uint32_t &y)
{
    uint32_t rx;
    const uint32_t ry = div_small(local, p.rect_w, rx);
    x = p.x0 + rx;
    y = p.y0 + ry;
    if (p.band_world > 1u) {
        uint32_t row_in_band;
        const uint32_t band = div_small(ry, p.band_rows, row_in_band);
        y = (band * p.band_world + p.band_rank) * p.band_rows + row_in_band;
    }
    return y * p.cam.width + x;
}

PT_DEV uint32_t frame_to_rect(const WavefrontParams &p, uint32_t pix)
{
    if (p.rect_w == p.cam.width && p.x0 == 0u && p.y0 == 0u && p.band_world <= 1u) return pix;   // wave-uniform
    uint32_t x;
    const uint32_t y = div_small(pix, p.cam.width, x);
    if (p.band_world > 1u) {
        uint32_t row_in_period;
        const uint32_t k = div_small(y, p.band_rows * p.band_world, row_in_period);
        return (k * p.band_rows + (row_in_period - p.band_rank * p.band_rows)) * p.rect_w + x;
    }
    return (y - p.y0) * p.rect_w + (x - p.x0);
}

// Index of a path within the batch: [iteration][pixel of the rectangle][sub-sample]
PT_DEV size_t batch_path_index(const WavefrontParams &p, uint32_t n_pix, uint32_t pix, uint32_t it_sub)
{
    return (((size_t)((it_sub >> 2) - p.first_iteration) * n_pix + frame_to_rect(p, pix)) << 2) | (it_sub & 3);
}

// parity hook: primitive index the path's Intersect call of this bounce returned
PT_DEV void record_hit(const WavefrontParams &p, uint32_t n_pix, uint32_t pix, uint32_t it_sub, int bounce, int best)
{
    if (p.hit_ids) p.hit_ids[batch_path_index(p, n_pix, pix, it_sub) * (size_t)(p.max_bounces + 1) + bounce] = best;
}

// Marks a path of the batch in the mask (byte path >> 2, bit path & 3: the path has an entry in
// path_radiance); returns whether it was marked before.
PT_DEV bool mark_path(const WavefrontParams &p, size_t path)
{
    const uint32_t bit = 1u << ((uint32_t)(path & 15u) >> 2 << 3 | (uint32_t)(path & 3u));
    return (atomicOr(p.path_mask + (path >> 4), bit) & bit) != 0u;
}

// Adds the emission a bounce collected to the path's radiance and clears it.  The entry takes the
// terms in bounce order, one launch (or one lane) after the other: the sum ((0 + e1) + e2) + ...
// is the one Trace builds in L (pathtracer.cu:136).  The buffer is never zeroed: a path's first
// term starts its entry and marks the path in the mask; one lane at a time works on a path.
PT_DEV void bank_radiance(const WavefrontParams &p, uint32_t n_pix, uint32_t pix, uint32_t it_sub, V3 &L)
{
    if (L.x != 0.f || L.y != 0.f || L.z != 0.f) {
        const size_t path = batch_path_index(p, n_pix, pix, it_sub);
        float *pr = p.path_radiance + path * 3;
        const V3 have = mark_path(p, path) ? mk(pr[0], pr[1], pr[2]) : mk(0.f, 0.f, 0.f);
        pr[0] = have.x + L.x; pr[1] = have.y + L.y; pr[2] = have.z + L.z;
        L = mk(0.f, 0.f, 0.f);
    }
}

template <bool TILED, bool BVH>
__global__ __launch_bounds__(kBlockThreads) void wf_step_kernel(const WavefrontParams p)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t q = blockIdx.x;                 // sub-queue
    const uint32_t n_pix = p.rect_w * p.rect_h;
    const uint64_t n_paths = (uint64_t)p.n_iterations * n_pix * 4;

    // Number of 256-path chunks of this sub-queue.  Bounce 0 has no input queue: path i of the
    // batch is (iteration offset, pixel, sub) in that order, and chunk c of sub-queue q is the
    // 256 paths starting at (c * 64 + q) * 256.
    uint32_t n_items = 0, n_chunks;
    if (p.bounce == 0) {
        const uint64_t blocks = (n_paths + kBlockThreads - 1) / kBlockThreads;
        n_chunks = (uint32_t)((blocks + kWfSubQueues - 1 - q) / kWfSubQueues);
    } else {
        n_items = *count_ptr(p.counts, p.bounce, q);
        n_chunks = (n_items + kBlockThreads - 1) / kBlockThreads;
    }
    if (blockIdx.y >= n_chunks) return;            // uniform for the whole workgroup

    const SceneView sc = stage_scene<TILED>(p.scene, lds_raw, tid);
    const uint32_t k0 = (uint32_t)p.seed, k1 = (uint32_t)(p.seed >> 32);

    for (uint32_t chunk = blockIdx.y; chunk < n_chunks; chunk += gridDim.y) {
        bool active;
        V3 o = mk(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
        V3 d = mk(0.f, 0.f, 1.f), T = mk(1.f, 1.f, 1.f), L = mk(0.f, 0.f, 0.f);
        uint32_t pix = 0, it_sub = 0;
        if (p.bounce == 0) {
            const uint64_t i = ((uint64_t)chunk * kWfSubQueues + q) * kBlockThreads + tid;
            active = i < n_paths;
            if (active) {
                const uint32_t sub = (uint32_t)(i & 3);
                const uint64_t ip = i >> 2;
                uint32_t local, x, y;
                const uint32_t it_off = div_small((uint32_t)ip, n_pix, local);   // a batch holds fewer than 2^32 paths
                pix = rect_to_frame(p, local, x, y);
                it_sub = ((p.first_iteration + it_off) << 2) | sub;
                d = camera_ray(p.cam, x, y, pix, it_sub >> 2, (int)sub, k0, k1);
            }
        } else {
            const uint32_t slot_in = chunk * kBlockThreads + tid;
            active = slot_in < n_items;
            if (active) {
                const size_t s = (size_t)q * p.sub_capacity + slot_in;
                const float4 a = p.in.a[s], bb = p.in.b[s];
                const WavefrontTail c = p.in.c[s];
                o = mk(a.x, a.y, a.z); d = mk(a.w, bb.x, bb.y); T = mk(bb.z, bb.w, c.tz);
                pix = c.pix; it_sub = c.it_sub;
            }
        }

        float tmax;
        int best;
        intersect_scene<TILED, BVH>(p.scene, sc, lds_raw, tid, active, o, d, tmax, best);

        bool cont = false;
        if (active) {
            bool end = true;
            int b = p.bounce;
            record_hit(p, n_pix, pix, it_sub, p.bounce, best);
            if (best >= 0) {
                const float4 s = TILED ? p.scene.sph[best] : sc.sph[best];
                end = shade_bounce(sc, s, best, tmax, pix, it_sub >> 2, k0, k1, p.max_bounces, o, d, T, L, b);
                bank_radiance(p, n_pix, pix, it_sub, L);
            }
            cont = !end;
        }

        // compaction: one atomic per wave reserves the survivors' slots in the same sub-queue
        const unsigned long long mask = __ballot(cont);
        if (mask) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const int leader = (int)__builtin_ctzll(mask);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(count_ptr(p.counts, p.bounce + 1, q), (uint32_t)__popcll(mask));
            base = __shfl(base, leader, 64);
            if (cont) {
                const size_t s = (size_t)q * p.sub_capacity + base + rank;
                p.out.a[s] = make_float4(o.x, o.y, o.z, d.x);
                p.out.b[s] = make_float4(d.y, d.z, T.x, T.y);
                p.out.c[s] = WavefrontTail{T.z, pix, it_sub};
            }
        }
    }
}

// BVH scenes: the same step with dynamic ray fetch.  A BVH traversal takes a different number of
// steps for every ray (mean 16 node and leaf visits, slowest lane of a wave above 100 on the
// 10k-sphere scene), so a wave that loads 64 paths and waits for the slowest runs at ~0.2 lane
// occupancy.  Here a wave is persistent: whenever fewer than p.dynamic_fetch of its lanes are still
// traversing, the finished lanes shade and store their paths and fetch new ones from the sub-queue
// (one atomic on the sub-queue's fetch cursor per refill), while the others keep their traversal
// state in registers and their stacks in LDS.  Results do not depend on which lane traces which
// path; only the order of the survivors in the next queue changes.
//
// WHOLE: the launch follows the paths it fetches (camera rays, or the survivors of bounce
// p.bounce - 1) to their end.  A lane whose path goes on keeps it and starts the next bounce's
// traversal at once; nothing is appended to a queue.
// The per-path radiance buffer and the resolve kernel make that possible -- no lane owns a pixel.
// No queue records move through memory (56 bytes written and read per path and bounce otherwise),
// and one launch has one tail of last, longest traversals instead of one per bounce.
//
// Measured and dropped: the tree's inner nodes staged in LDS by one 1024-thread workgroup per CU
// (4-byte stack entries, 130 KiB of LDS on the 10 007-sphere scene) -- 61.3 ms per 8 iterations
// against 53.8 ms: four waves per SIMD instead of six cost more than the shorter fetches gain.
//
// NEE (with WHOLE, from the camera rays on): the direct-light integrator.  The ray a lane traverses
// is its path ray or, in the pass after a diffuse vertex parked a direct-light term, the shadow ray
// of that vertex (`rd`, bounded by the distance to a point light).
template <bool WHOLE, bool NEE = false>
__global__ __launch_bounds__(kBlockThreads, NEE ? PT_NEE_BVH_WAVES : PT_BVH_WAVES) void wf_step_bvh_kernel(const WavefrontParams p)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr int BT = kBlockThreads;
    const int tid = threadIdx.x;
    const uint32_t lane = (uint32_t)tid & 63u;
    const uint32_t q = blockIdx.x;
    const uint32_t n_pix = p.rect_w * p.rect_h;
    const uint64_t n_paths = (uint64_t)p.n_iterations * n_pix * 4;

    // items of this sub-queue: item j is slot j of the input queue, or at bounce 0 path
    // ((j / 256) * 64 + q) * 256 + j % 256 of the batch (the static kernel's assignment: a block of
    // 256 paths is 64 pixels of an image row; 8x8-pixel tiles instead measured 4 % slower)
    uint32_t n_items;
    if (p.bounce == 0) {
        const uint64_t blocks = (n_paths + kBlockThreads - 1) / kBlockThreads;
        n_items = (uint32_t)((blocks + kWfSubQueues - 1 - q) / kWfSubQueues) * kBlockThreads;
    } else {
        n_items = *count_ptr(p.counts, p.bounce, q);
    }
    if (blockIdx.y * (uint32_t)BT >= n_items) return;
    uint32_t *cursor = count_ptr(p.counts, p.bounce, q) + 1;   // second word of the counter's line

    const BvhParams &B = p.scene.bvh;
    const SceneView sc = stage_scene<true>(p.scene, lds_raw, tid);
    BvhStack8<BT> stack;   // LDS: [stack entries][BT lanes]
    stack.base = reinterpret_cast<uint32_t *>(lds_raw) + tid;
    const uint4 *nodes = B.nodes;
    const uint32_t k0 = (uint32_t)p.seed, k1 = (uint32_t)(p.seed >> 32);

    bool has = false, trav = false, over = false, exhausted = false;
    bool start = false;                 // the lane has a new ray whose traversal is not set up yet
    V3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f), T = mk(1.f, 1.f, 1.f), L = mk(0.f, 0.f, 0.f);
    uint32_t pix = 0, it_sub = 0;
    int bounce = p.bounce;              // WHOLE: per lane
    unsigned long long continued = 0;   // WHOLE: Intersect calls beyond the camera rays (wave-uniform)
    float tmax = PT_INF;
    int best = -1;
    BvhTraversal tr;
    tr.cur = 0; tr.sp = 0; tr.visits = 0;
    NeeState st;
    st.shadow = false; st.end_after = false; st.specular_bounce = false; st.bsdf_pdf_prev = 0.f;
    st.sh_d = mk(0.f, 0.f, 1.f); st.sh_tmax = PT_INF; st.sh_prim = -1; st.pending = mk(0.f, 0.f, 0.f);
    V3 rd = d;                          // NEE: the direction being traversed (d or the shadow ray's)

    for (;;) {
        // ---- refill idle lanes ---------------------------------------------------------------
        if (!exhausted) {
            const unsigned long long idle = __ballot(!has);
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            // n_idle > 0 here: the traversal loop below only ends with finished lanes
            const int leader = (int)__builtin_ctzll(idle | (1ull << 63));
            uint32_t base = 0;
            if ((int)lane == leader && n_idle) base = atomicAdd(cursor, n_idle);
            base = __shfl(base, leader, 64);
            if (base + n_idle >= n_items) exhausted = true;
            const uint32_t j = base + rank;
            if (!has && j < n_items) {
                if (p.bounce == 0) {
                    const uint64_t i = ((uint64_t)(j >> 8) * kWfSubQueues + q) * kBlockThreads + (j & 255u);
                    if (i < n_paths) {
                        const uint32_t sub = (uint32_t)(i & 3);
                        const uint64_t ip = i >> 2;
                        uint32_t local, x, y;
                        const uint32_t it_off = div_small((uint32_t)ip, n_pix, local);
                        pix = rect_to_frame(p, local, x, y);
                        it_sub = ((p.first_iteration + it_off) << 2) | sub;
                        o = mk(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
                        d = camera_ray(p.cam, x, y, pix, it_sub >> 2, (int)sub, k0, k1);
                        T = mk(1.f, 1.f, 1.f);
                        L = mk(0.f, 0.f, 0.f);
                        has = true;
                    }
                } else {
                    const size_t s = (size_t)q * p.sub_capacity + j;
                    const float4 a = p.in.a[s], bb = p.in.b[s];
                    const WavefrontTail c = p.in.c[s];
                    o = mk(a.x, a.y, a.z); d = mk(a.w, bb.x, bb.y); T = mk(bb.z, bb.w, c.tz);
                    L = mk(0.f, 0.f, 0.f);
                    pix = c.pix; it_sub = c.it_sub;
                    has = true;
                }
                if (has) {
                    if (WHOLE) bounce = p.bounce;
                    if (NEE) { st.shadow = false; st.specular_bounce = false; }
                    start = true;
                }
            }
        }
        if (!__any(has)) break;
        // One set-up (the always-tested spheres, the ray's slab constants) for the fetched rays and
        // for the paths that continue in their lane: two call sites would run one after the other.
        if (start) {
            start = false;
            tmax = (NEE && st.shadow) ? st.sh_tmax : PT_INF;
            best = -1;
            over = false;
            if (NEE) rd = st.shadow ? st.sh_d : d;
            trav = bvh_begin(B, o, NEE ? rd : d, tmax, best, tr);
        }

        // ---- traverse until too few lanes are busy ---------------------------------------------
        for (;;) {
            // A fixed number of box steps per leaf step: waiting until no lane of the wave is at an
            // inner node any more (the classic while-while) measured 25 % slower, one step 8 %.
#pragma unroll 1
            for (uint32_t k = 0; k < B.node_steps; ++k) {
                if (trav && tr.cur >= 0) {
                    trav = bvh_node_step(B, nodes, stack, tmax, tr);
                    if (tr.visits > B.budget) { over = trav; trav = false; }
                }
            }
            if (trav && tr.cur < 0) trav = bvh_leaf_step(B, stack, o, NEE ? rd : d, tmax, best, tr);
            const uint32_t busy = (uint32_t)__popcll(__ballot(trav));
            if (busy == 0 || (!exhausted && busy < (uint32_t)p.dynamic_fetch)) break;
        }

        // ---- lanes whose traversal is finished: shade, store, free the lane ---------------------
        const bool fin = has && !trav;
        PT_LS(4);
        if (fin) PT_LS(17);
        if (fin) best = bvh_sphere_index(B, best);
        bvh_finish_over_budget(p.scene.sph, p.scene.n_prims, fin && over, o, NEE ? rd : d, tmax, best);
        bool cont = false;
        if (fin) {
            bool end = true;
            int b = WHOLE ? bounce : p.bounce;
            if (NEE && st.shadow) {
                // the shadow query's answer: an area light is visible when the nearest hit is its own sphere,
                // a point light when nothing lies within the segment
                const bool visible = st.sh_prim >= 0 ? best == st.sh_prim : best < 0;
                if (visible) L = L + st.pending;
                st.shadow = false;
                end = st.end_after;
            } else {
                record_hit(p, n_pix, pix, it_sub, b, best);
                if (best >= 0) {
                    if (NEE) {
                        end = shade_bounce_nee(p.scene, sc, p.scene.sph[best], best, tmax, pix, it_sub >> 2, k0, k1, p.max_bounces, o, d, T, L, b, st);
                        if (st.shadow) { st.end_after = end; end = false; }
                    } else {
                        end = shade_bounce(sc, p.scene.sph[best], best, tmax, pix, it_sub >> 2, k0, k1, p.max_bounces, o, d, T, L, b);
                    }
                }
            }
            // NEE (whole paths only): the lane keeps L, emission and direct-light terms in the order they
            // arise, and starts the path's entry with it when the path ends
            if (!NEE || end) bank_radiance(p, n_pix, pix, it_sub, L);
            cont = !end;
            has = false;
            if (WHOLE && cont) {   // the same lane traces the path's next ray
                bounce = b;
                has = true;
                start = true;
            }
        }
        const unsigned long long mask = __ballot(cont);
        if (WHOLE) continued += (unsigned long long)__popcll(mask);
        if (!WHOLE && mask) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const int leader = (int)__builtin_ctzll(mask);
            uint32_t base = 0;
            if ((int)lane == leader) base = atomicAdd(count_ptr(p.counts, p.bounce + 1, q), (uint32_t)__popcll(mask));
            base = __shfl(base, leader, 64);
            if (cont) {
                const size_t s = (size_t)q * p.sub_capacity + base + rank;
                p.out.a[s] = make_float4(o.x, o.y, o.z, d.x);
                p.out.b[s] = make_float4(d.y, d.z, T.x, T.y);
                p.out.c[s] = WavefrontTail{T.z, pix, it_sub};
            }
        }
        if (exhausted && !__any(has)) break;
    }
    // the resolve kernel counts the camera rays (and, per bounce, the queue lengths: zero here)
    if (WHOLE && p.ray_counter && continued && lane == 0) atomicAdd(p.ray_counter, continued);
}

// Path pool for resident scenes (linear scan, a few iterations per launch): persistent waves, no
// lane owns a pixel.  A launch of one iteration through the megakernel gives every lane the four
// paths of its pixel; the lanes of a wave need different numbers of bounces for them, and what the
// early ones wait is a fifth of the launch (lanes with a ray 0.79), on top of 8160 workgroup
// start-ups for four paths per lane.  Here a lane whose path ended takes the next camera ray of
// its wave's pool, whatever pixel it belongs to; the path's radiance goes to the per-path buffer
// when the path ends and wf_resolve folds the buffer into the frame in the reference's order.  The pool: the wave
// generates 64 camera rays at a time with all lanes active (paths of the batch in the order of
// the static kernel's bounce 0, 128 consecutive items of the workgroup's sub-queue per refill: every
// lane two neighbouring sub-samples of a pixel-iteration, which share a Philox block -- 2.5 % faster
// than 64 rays from 64 blocks) into a ring of 192 entries in LDS (up to 63 waiting + one refill),
// idle lanes pop.
//
// TILED: scenes of more than 512 spheres rendered without the BVH, whose spheres pass through LDS
// one tile at a time.  The tile loads are workgroup barriers, so the four waves make their passes
// together: a wave whose pool has run dry keeps loading tiles until the other three are done too.
// Camera rays are a negligible part of the work there and LDS decides how many workgroups a CU
// holds: one ray per lane and refill, ring of 128.
constexpr uint32_t pool_ring(bool tiled) { return tiled ? 128u : 192u; }
struct PoolEntry { float dx, dy, dz; uint32_t pix, it_sub; };

//
// NEE: the direct-light integrator (shade_bounce_nee).  A diffuse vertex parks its direct-light
// term and the lane's next pass traces the shadow ray, as in render_nee_kernel; path rays and shadow
// rays of different lanes share the scan.  Nothing else changes: the path still lives in one lane
// from its camera ray to its end and stores its radiance once.
// NEE_MODE: 0 the path integrator, 1 the direct-light integrator with a shadow pass, 2 with a second scan
// (see the loop body).
// TRI: the scene has triangles behind its spheres (path integrator only): they continue the scan, and a hit on
// one is shaded with the triangle's own normal.
template <bool TILED, int NEE_MODE, bool TRI = false>
__global__ __launch_bounds__(kBlockThreads, NEE_MODE == 2 ? PT_NEE_POOL_WAVES : NEE_MODE == 1 ? PT_NEE_SHADOW_PASS_WAVES : 1) void wf_pool_kernel(const WavefrontParams p)
{
    static_assert(!(TRI && NEE_MODE != 0), "the direct-light integrator samples spheres only");
    constexpr bool NEE = NEE_MODE != 0;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int tid = threadIdx.x;
    const uint32_t lane = (uint32_t)tid & 63u;
    const uint32_t q = blockIdx.x;
    const uint32_t n_pix = p.rect_w * p.rect_h;
    const uint64_t n_paths = (uint64_t)p.n_iterations * n_pix * 4;
    const uint64_t blocks = (n_paths + kBlockThreads - 1) / kBlockThreads;
    const uint32_t n_items = (uint32_t)((blocks + kWfSubQueues - 1 - q) / kWfSubQueues) * kBlockThreads;
    if (blockIdx.y * (uint32_t)kBlockThreads >= n_items) return;   // uniform for the workgroup
    uint32_t *cursor = count_ptr(p.counts, 0, q) + 1;

    const SceneView sc = stage_scene<TILED>(p.scene, lds_raw, tid);
    const uint32_t scene_bytes = TILED ? kTileSpheres * (uint32_t)sizeof(float4) : scene_lds_bytes_dev(p.scene);
    constexpr uint32_t kRing = pool_ring(TILED);
    constexpr uint32_t kRefill = TILED ? 64u : 128u;   // items per refill
    auto wrap = [](uint32_t i) { return i >= kRing ? i - kRing : i; };
    PoolEntry *ring = reinterpret_cast<PoolEntry *>(lds_raw + scene_bytes) + (tid >> 6) * kRing;
    const uint32_t k0 = (uint32_t)p.seed, k1 = (uint32_t)(p.seed >> 32);

    uint32_t head = 0, count = 0;          // the wave's ring (wave-uniform)
    bool exhausted = false;                // the sub-queue has no more items
    unsigned long long continued = 0;      // Intersect calls beyond the camera rays (wave-uniform)
    bool has = false;
    V3 o = mk(0.f, 0.f, 0.f), d = mk(0.f, 0.f, 1.f), T = mk(1.f, 1.f, 1.f), L = mk(0.f, 0.f, 0.f);
    uint32_t pix = 0, it_sub = 0;
    int b = 0;
    NeeState st;
    st.shadow = false; st.end_after = false; st.specular_bounce = false; st.bsdf_pdf_prev = 0.f;
    st.sh_d = mk(0.f, 0.f, 1.f); st.sh_tmax = PT_INF; st.sh_prim = -1; st.pending = mk(0.f, 0.f, 0.f);

    for (;;) {
        const unsigned long long idle = __ballot(!has);
        const uint32_t n_idle = (uint32_t)__popcll(idle);
        if (n_idle) {
            if (count < n_idle && !exhausted) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(cursor, kRefill);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (base + kRefill >= n_items) exhausted = true;
                // TILED: item j.  Otherwise items j and j + 1: sub-samples (0, 1) or (2, 3) of a pixel-iteration
                const uint32_t j = base + (TILED ? lane : 2u * lane);
                const uint64_t i = ((uint64_t)(j >> 8) * kWfSubQueues + q) * kBlockThreads + (j & 255u);
                const bool valid = j < n_items && i < n_paths;   // the number of paths is a multiple of 4
                PoolEntry e0, e1;   // written to the ring by valid lanes only
                if (valid) {
                    const uint64_t ip = i >> 2;
                    uint32_t local, x, y;
                    const uint32_t it_off = div_small((uint32_t)ip, n_pix, local);
                    e0.pix = e1.pix = rect_to_frame(p, local, x, y);
                    e0.it_sub = ((p.first_iteration + it_off) << 2) | (uint32_t)(i & 3);
                    e1.it_sub = e0.it_sub | 1u;
                    V3 dirs[2];
                    if (TILED) {
                        dirs[0] = camera_ray(p.cam, x, y, e0.pix, e0.it_sub >> 2, (int)(i & 3), k0, k1);
                        dirs[1] = dirs[0];
                    } else {
                        camera_ray_pair(p.cam, x, y, e0.pix, e0.it_sub >> 2, (uint32_t)(i >> 1) & 1u, k0, k1, dirs);
                    }
                    e0.dx = dirs[0].x; e0.dy = dirs[0].y; e0.dz = dirs[0].z;
                    e1.dx = dirs[1].x; e1.dy = dirs[1].y; e1.dz = dirs[1].z;
                }
                const unsigned long long vmask = __ballot(valid);
                const uint32_t vrank = __builtin_amdgcn_mbcnt_hi((uint32_t)(vmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)vmask, 0u));
                if (valid) {
                    const uint32_t slot = wrap(head + count + (TILED ? vrank : 2u * vrank));
                    ring[slot] = e0;
                    if (!TILED) ring[wrap(slot + 1u)] = e1;
                }
                count += (TILED ? 1u : 2u) * (uint32_t)__popcll(vmask);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            if (!has && rank < count) {
                const PoolEntry e = ring[wrap(head + rank)];
                o = mk(p.cam.pos[0], p.cam.pos[1], p.cam.pos[2]);
                d = mk(e.dx, e.dy, e.dz);
                T = mk(1.f, 1.f, 1.f);
                L = mk(0.f, 0.f, 0.f);
                pix = e.pix; it_sub = e.it_sub;
                b = 0;
                if (NEE) { st.shadow = false; st.specular_bounce = false; }
                has = true;
            }
            const uint32_t taken = n_idle < count ? n_idle : count;
            head = wrap(head + taken);
            count -= taken;
            __builtin_amdgcn_wave_barrier();
        }
        if (TILED) {
            // all four waves leave together (barriers in the scan)
            if (__syncthreads_and(!__any(has) && exhausted && count == 0)) break;
        } else if (!__any(has)) {
            // nothing to trace: the pool has run dry, or a refill brought no valid item yet (the
            // batch's last, partly empty block)
            if (exhausted && count == 0) break;
        }
        if (TILED || __any(has)) {
            float tmax;
            int best;
            PT_LS(4);
            if (has) PT_LS(17);
            // NEE with a shadow pass: a pass is ONE scan, which path rays and the shadow rays of the vertices
            // shaded in the pass before share -- for scenes whose scan is what a pass costs.  NEE with a second
            // scan: the shadow ray is traced in the pass that shaded its vertex; a lane that only waited for its
            // shadow ray's answer would idle through the shading code of the others, which on a scene of a few
            // spheres costs more than scanning them twice (1080p, demo scene of 9 spheres: 64.2 -> 57.7 ms per 64
            // iterations; 71 spheres at depth 16: 57.5 -> 66.4 ms per 32).
            constexpr bool kShadowPass = NEE_MODE == 1;
            const bool shadow = kShadowPass && has && st.shadow;   // this pass answers the shadow query of the vertex shaded last
            if (kShadowPass) intersect_scene<TILED, false>(p.scene, sc, lds_raw, tid, has, o, shadow ? st.sh_d : d, tmax, best, shadow ? st.sh_tmax : PT_INF);
            else             intersect_scene<TILED, false, TRI>(p.scene, sc, lds_raw, tid, has, o, d, tmax, best);
            bool cont = false, end = true;
            // without a shadow pass the shadow ray's fields live inside one pass only: a pass-local copy,
            // so that they hold no registers across the loop
            NeeState pass_state;
            if (NEE && !kShadowPass) {
                pass_state.shadow = false; pass_state.end_after = false;
                pass_state.specular_bounce = st.specular_bounce; pass_state.bsdf_pdf_prev = st.bsdf_pdf_prev;
                pass_state.sh_d = mk(0.f, 0.f, 1.f); pass_state.sh_tmax = PT_INF; pass_state.sh_prim = -1; pass_state.pending = mk(0.f, 0.f, 0.f);
            }
            NeeState &ns = kShadowPass ? st : pass_state;
            if (has) {
                if (kShadowPass && shadow) {
                    const bool visible = st.sh_prim >= 0 ? best == st.sh_prim : best < 0;
                    if (visible) L = L + st.pending;
                    st.shadow = false;
                    end = st.end_after;
                } else if (NEE) {
                    record_hit(p, n_pix, pix, it_sub, b, best);
                    if (best >= 0) {
                        end = shade_bounce_nee(p.scene, sc, (TILED ? p.scene.sph : sc.sph)[best], best, tmax, pix, it_sub >> 2, k0, k1, p.max_bounces, o, d, T, L, b, ns);
                        if (kShadowPass && st.shadow) { st.end_after = end; end = false; }
                    }
                } else {
                    record_hit(p, n_pix, pix, it_sub, b, best);
                    if (TRI && best >= (int)p.scene.n_prims) {
                        const float4 *tri4 = (TILED ? p.scene.tri : sc.tri) + 4 * (size_t)(best - (int)p.scene.n_prims);
                        end = shade_bounce_triangle(sc, tri4, tmax, pix, it_sub >> 2, k0, k1, p.max_bounces, o, d, T, L, b);
                    } else if (best >= 0) {
                        end = shade_bounce(sc, (TILED ? p.scene.sph : sc.sph)[best], best, tmax, pix, it_sub >> 2, k0, k1, p.max_bounces, o, d, T, L, b);
                    }
                }
            }
            if (NEE && !kShadowPass) {
                // the shadow rays of the vertices just shaded (o is the vertex): an area light is visible when
                // the nearest hit is its own sphere, a point light when nothing lies within the segment
                const bool sh = has && pass_state.shadow;
                st.specular_bounce = pass_state.specular_bounce; st.bsdf_pdf_prev = pass_state.bsdf_pdf_prev;
                float sh_t;
                int sh_best;
                intersect_scene<TILED, false>(p.scene, sc, lds_raw, tid, sh, o, pass_state.sh_d, sh_t, sh_best, pass_state.sh_tmax);
                if (sh) {
                    const bool visible = pass_state.sh_prim >= 0 ? sh_best == pass_state.sh_prim : sh_best < 0;
                    if (visible) L = L + pass_state.pending;
                }
                continued += (unsigned long long)__popcll(__ballot(sh));   // an Intersect call like any other
            }
            if (has) {
                // The path lived in this lane from its camera ray on: one store when it ends, and only
                // if it met a light -- most paths end black, and adding their +0 changes no sum.
                if (end && (L.x != 0.f || L.y != 0.f || L.z != 0.f || p.store_all_paths)) {
                    const size_t path = batch_path_index(p, n_pix, pix, it_sub);
                    float *pr = p.path_radiance + path * 3;
                    pr[0] = L.x; pr[1] = L.y; pr[2] = L.z;
                    (void)mark_path(p, path);
                }
                cont = !end;
                has = cont;
            }
            continued += (unsigned long long)__popcll(__ballot(cont));
        }
    }
    // the resolve kernel counts the camera rays
    if (p.ray_counter && continued && lane == 0) atomicAdd(p.ray_counter, continued);
}

// Folds the batch's path radiances into the frame buffer: per pixel, iteration by iteration,
// c = (((0+L0)+L1)+L2)+L3) * 1/4 and the running mean / sum of RenderKernel (pathtracer.cu:94-109).
__global__ __launch_bounds__(kBlockThreads) void wf_resolve_kernel(const WavefrontParams p)
{
    const uint32_t n_pix = p.rect_w * p.rect_h;
    const uint32_t local = blockIdx.x * kBlockThreads + threadIdx.x;
    if (local < n_pix) {
        uint32_t x, y;
        float *px = p.buffer ? p.buffer + (size_t)rect_to_frame(p, local, x, y) * 3 : nullptr;
        V3 acc = (p.overwrite || !px) ? mk(0.f, 0.f, 0.f) : mk(px[0], px[1], px[2]);
        V3 cpix = mk(0.f, 0.f, 0.f);
        for (uint32_t k = 0; k < p.n_iterations; ++k) {
            const size_t group = (size_t)k * n_pix + local;
            const float *pr = p.path_radiance + (group << 2) * 3;
            V3 c = mk(0.f, 0.f, 0.f);
            // only the marked sub-samples have an entry; the others were black (c + 0 == c)
            unsigned char *mask = reinterpret_cast<unsigned char *>(p.path_mask) + group;
            const uint32_t m = *mask;
            if (m) {
                *mask = 0;   // for the buffer set's next batch
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    if (m & (1u << s)) c = c + mk(pr[s * 3], pr[s * 3 + 1], pr[s * 3 + 2]);
            }
            acc = accumulate_pixel(acc, c, p.first_iteration + k, p.accum_mode, &cpix);
        }
        if (px) { px[0] = acc.x; px[1] = acc.y; px[2] = acc.z; }
        if (p.pixel_rgb) {   // parity hook: the colour of the batch's last iteration
            float *pr = p.pixel_rgb + (size_t)local * 3;
            pr[0] = cpix.x; pr[1] = cpix.y; pr[2] = cpix.z;
        }
    }
    // Intersect calls of the batch = paths entering each bounce.  The counters and cursors are left
    // at zero for the buffer set's next batch (every launch of this batch has finished: stream order).
    if (blockIdx.x == 0 && threadIdx.x < kWfSubQueues) {
        unsigned long long n = 0;
        for (int b = 0; b <= p.max_bounces + 1; ++b) {
            uint32_t *c = count_ptr(p.counts, b, threadIdx.x);
            if (b >= 1 && b <= p.max_bounces) n += c[0];
            c[0] = 0u; c[1] = 0u;
        }
        if (threadIdx.x == 0) n += (unsigned long long)p.n_iterations * n_pix * 4;
        if (p.ray_counter) atomicAdd(p.ray_counter, n);
    }
}

} // namespace

// Divergence counters of the kernels in this file (measurement builds only), added to out[64] and cleared.
hipError_t add_lane_stats_wavefront(unsigned long long out[64])
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

// Compute units of the current device (looked up once per device; 256 if the query fails).
static int current_device_cus()
{
    constexpr int kMaxDevices = 64;
    static std::atomic<int> cached[kMaxDevices];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

hipError_t launch_wavefront_step(const WavefrontParams &p, bool tiled, hipStream_t stream)
{
    // 64 sub-queues x up to 32 workgroups each (2048 workgroups = one resident set of the chip);
    // a workgroup walks its sub-queue's chunks with stride gridDim.y and returns at once when the
    // sub-queue is shorter than its first chunk.
    const uint32_t chunks = p.sub_capacity / kBlockThreads;
    const dim3 grid(kWfSubQueues, chunks < 32u ? chunks : 32u), block(kBlockThreads);
    const bool bvh = p.scene.bvh.n_nodes > 0;   // BVH kernels read the scene from global memory
    if (p.path_pool && !bvh) {
        // one resident set of persistent workgroups: 8 per CU (8 waves per SIMD), fewer if LDS says so
        const size_t pool_lds = scene_lds_bytes(p.scene, tiled) + (kBlockThreads / 64) * pool_ring(tiled) * sizeof(PoolEntry);
        uint32_t per_cu = (uint32_t)((160u * 1024u) / pool_lds);
        if (per_cu > 8u) per_cu = 8u;
        // the direct-light integrator's registers
        const uint32_t nee_waves = (tiled || p.scene.n_prims > kMaxRingSpheres) ? PT_NEE_SHADOW_PASS_WAVES : PT_NEE_POOL_WAVES;
        if (p.nee && per_cu > nee_waves) per_cu = nee_waves;
        if (p.pool_workgroups_per_cu > 0 && per_cu > (uint32_t)p.pool_workgroups_per_cu) per_cu = (uint32_t)p.pool_workgroups_per_cu;
        if (per_cu < 1u) per_cu = 1u;
        const uint32_t rows = (per_cu * (uint32_t)current_device_cus() + kWfSubQueues - 1) / kWfSubQueues;
        const uint32_t pool_chunks = p.sub_capacity / kBlockThreads;
        const dim3 pool_grid(kWfSubQueues, pool_chunks < rows ? pool_chunks : rows);
        if (p.nee) {
            if (tiled) hipLaunchKernelGGL((wf_pool_kernel<true, 1>), pool_grid, block, pool_lds, stream, p);
            else if (p.scene.n_prims > kMaxRingSpheres) hipLaunchKernelGGL((wf_pool_kernel<false, 1>), pool_grid, block, pool_lds, stream, p);
            else       hipLaunchKernelGGL((wf_pool_kernel<false, 2>), pool_grid, block, pool_lds, stream, p);
        } else if (p.scene.n_tris) {
            if (tiled) hipLaunchKernelGGL((wf_pool_kernel<true, 0, true>), pool_grid, block, pool_lds, stream, p);
            else       hipLaunchKernelGGL((wf_pool_kernel<false, 0, true>), pool_grid, block, pool_lds, stream, p);
        } else if (tiled) hipLaunchKernelGGL((wf_pool_kernel<true, 0>), pool_grid, block, pool_lds, stream, p);
        else              hipLaunchKernelGGL((wf_pool_kernel<false, 0>), pool_grid, block, pool_lds, stream, p);
        return hipGetLastError();
    }
    const size_t lds = bvh ? bvh_stack_lds_bytes(p.scene.bvh) : scene_lds_bytes(p.scene, tiled);
    if (bvh && p.dynamic_fetch) {
        // One resident set of persistent workgroups: per CU as many as the traversal stacks in LDS
        // (2 KiB per tree level) and the kernel's registers (6 waves per SIMD) allow.  More would
        // run as a second round behind the first and wait for it.
        const int n_cus = current_device_cus();
        uint32_t per_cu = (uint32_t)((160u * 1024u) / (lds ? lds : 1));
        const uint32_t by_registers = (p.whole_paths && p.nee) ? PT_NEE_BVH_WAVES : (uint32_t)PT_BVH_WAVES;
        if (per_cu > by_registers) per_cu = by_registers;
        if (per_cu < 1u) per_cu = 1u;
        uint32_t rows = (per_cu * (uint32_t)n_cus + kWfSubQueues - 1) / kWfSubQueues;
        const dim3 pgrid(kWfSubQueues, chunks < rows ? chunks : rows);
        if (p.whole_paths && p.nee) hipLaunchKernelGGL((wf_step_bvh_kernel<true, true>), pgrid, block, lds, stream, p);
        else if (p.whole_paths)     hipLaunchKernelGGL((wf_step_bvh_kernel<true, false>), pgrid, block, lds, stream, p);
        else                        hipLaunchKernelGGL((wf_step_bvh_kernel<false, false>), pgrid, block, lds, stream, p);
    } else if (bvh) hipLaunchKernelGGL((wf_step_kernel<true, true>), grid, block, lds, stream, p);
    else if (tiled) hipLaunchKernelGGL((wf_step_kernel<true, false>), grid, block, lds, stream, p);
    else            hipLaunchKernelGGL((wf_step_kernel<false, false>), grid, block, lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_wavefront_resolve(const WavefrontParams &p, hipStream_t stream)
{
    const uint32_t n_pix = p.rect_w * p.rect_h;
    hipLaunchKernelGGL(wf_resolve_kernel, dim3((n_pix + kBlockThreads - 1) / kBlockThreads), dim3(kBlockThreads), 0, stream, p);
    return hipGetLastError();
}

} // namespace ptmi
