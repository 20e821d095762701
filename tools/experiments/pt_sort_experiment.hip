// MEASUREMENT ONLY, not part of the product library (built by tools/build_sort_experiment.sh with
// -DPT_SORT_EXPERIMENT): sorts the survivors of a bounce inside every sub-queue before the next
// bounce's launch, with a device-wide segmented radix sort, to measure how much faster the BVH
// traversal runs on coherent rays -- the upper bound of what any binning of the queues can gain.
#include "pt_trace.hpp"

#include <cstring>
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace ptmi {

namespace {

struct SortScratch {
    uint32_t *keys[2] = {nullptr, nullptr}, *vals[2] = {nullptr, nullptr}, *begin = nullptr, *end = nullptr;
    void *tmp = nullptr;
    size_t tmp_bytes = 0, slots = 0;
} g_sort;

__device__ uint32_t spread3(uint32_t v)   // 5 bits -> every third bit
{
    uint32_t r = 0;
    for (int i = 0; i < 5; ++i) r |= ((v >> i) & 1u) << (3 * i);
    return r;
}

__global__ void sort_keys_kernel(WavefrontQueue q, const uint32_t *counts, uint32_t sub_capacity, int mode, float3 lo, float3 inv_cell,
                                 uint32_t *keys, uint32_t *vals, uint32_t *begin, uint32_t *end)
{
    const uint32_t sub = blockIdx.y;
    const uint32_t n = counts[(size_t)sub * kWfCountStride];
    if (blockIdx.x == 0 && threadIdx.x == 0) { begin[sub] = sub * sub_capacity; end[sub] = sub * sub_capacity + n; }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t s = (size_t)sub * sub_capacity + i;
        const float4 a = q.a[s], b = q.b[s];
        const uint32_t oct = (a.w < 0.f ? 1u : 0u) | (b.x < 0.f ? 2u : 0u) | (b.y < 0.f ? 4u : 0u);
        auto cell = [](float x, float l, float inv) { const float c = (x - l) * inv; return (uint32_t)(c < 0.f ? 0.f : (c > 31.f ? 31.f : c)); };
        const uint32_t morton = spread3(cell(a.x, lo.x, inv_cell.x)) | (spread3(cell(a.y, lo.y, inv_cell.y)) << 1) | (spread3(cell(a.z, lo.z, inv_cell.z)) << 2);
        uint32_t key = 0;
        if (mode == 1) key = oct;
        else if (mode == 2) key = (oct << 15) | morton;
        else if (mode == 3) key = (morton << 3) | oct;
        else if (mode == 4) key = ((morton >> 9) << 3) | oct;   // 4 x 4 x 4 cells, then the octant
        keys[s] = key;
        vals[s] = (uint32_t)s;
    }
}

__global__ void sort_gather_kernel(WavefrontQueue from, WavefrontQueue to, const uint32_t *counts, uint32_t sub_capacity, const uint32_t *vals)
{
    const uint32_t sub = blockIdx.y;
    const uint32_t n = counts[(size_t)sub * kWfCountStride];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const size_t s = (size_t)sub * sub_capacity + i;
        const uint32_t src = vals[s];
        to.a[s] = from.a[src]; to.b[s] = from.b[src]; to.c[s] = from.c[src];
    }
}

} // namespace

// Sorts the records of `from` (the queue a bounce's launch has just filled; counts = its per-sub-queue
// counters) into `to`, sub-queue by sub-queue.  mode: 1 octant, 2 octant then origin cell, 3 origin
// cell then octant, 4 coarse origin cell then octant.
hipError_t sort_queue_experiment(const WavefrontParams &p, WavefrontQueue from, WavefrontQueue to, const uint32_t *counts, int mode,
                                 hipStream_t stream)
{
    const size_t slots = (size_t)p.sub_capacity * kWfSubQueues;
    if (slots > g_sort.slots) {
        for (int k = 0; k < 2; ++k) {
            if (hipMalloc((void **)&g_sort.keys[k], slots * 4) != hipSuccess || hipMalloc((void **)&g_sort.vals[k], slots * 4) != hipSuccess)
                return hipErrorOutOfMemory;
        }
        if (!g_sort.begin && (hipMalloc((void **)&g_sort.begin, 256) != hipSuccess || hipMalloc((void **)&g_sort.end, 256) != hipSuccess))
            return hipErrorOutOfMemory;
        g_sort.slots = slots;
    }
    const BvhParams &B = p.scene.bvh;
    const float3 lo = make_float3(B.centre[0] - B.half_diag, B.centre[1] - B.half_diag, B.centre[2] - B.half_diag);
    const float inv = 32.f / (2.f * B.half_diag);
    hipLaunchKernelGGL(sort_keys_kernel, dim3(256, kWfSubQueues), dim3(256), 0, stream, from, counts, p.sub_capacity, mode, lo,
                       make_float3(inv, inv, inv), g_sort.keys[0], g_sort.vals[0], g_sort.begin, g_sort.end);
    size_t need = 0;
    hipError_t e = rocprim::segmented_radix_sort_pairs(nullptr, need, g_sort.keys[0], g_sort.keys[1], g_sort.vals[0], g_sort.vals[1],
                                                       (unsigned int)slots, kWfSubQueues, g_sort.begin, g_sort.end, 0, 32, stream);
    if (e != hipSuccess) return e;
    if (need > g_sort.tmp_bytes) {
        if (g_sort.tmp) (void)hipFree(g_sort.tmp);
        if (hipMalloc(&g_sort.tmp, need) != hipSuccess) return hipErrorOutOfMemory;
        g_sort.tmp_bytes = need;
    }
    e = rocprim::segmented_radix_sort_pairs(g_sort.tmp, need, g_sort.keys[0], g_sort.keys[1], g_sort.vals[0], g_sort.vals[1],
                                            (unsigned int)slots, kWfSubQueues, g_sort.begin, g_sort.end, 0, 32, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sort_gather_kernel, dim3(256, kWfSubQueues), dim3(256), 0, stream, from, to, counts, p.sub_capacity, g_sort.vals[1]);
    return hipGetLastError();
}

} // namespace ptmi
