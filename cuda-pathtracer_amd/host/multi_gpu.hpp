// One render sharded over the GPUs of a node, from ONE process: a host thread and a Pathtracer
// instance per device, no exchange while rendering, and one RCCL reduce(sum) of the accumulation
// buffers into device 0 at the end.  The reference is single-GPU (main.cpp:386-409 creates one
// Pathtracer on the current device); this is what replaces its `main` on a multi-GPU node.
//
// Two ways to cut the work (every (pixel, iteration, sub-sample) is independent given the
// counter-based random numbers):
//   tile split    the frame is cut into bands of 16 rows, band b belongs to device b % world;
//                 a device renders ALL iterations of its bands as a running mean and leaves the
//                 other pixels at zero, so the sum over devices is the one-GPU image bit for bit;
//   sample split  device r renders iterations [first_r, first_r + n_r) of the WHOLE frame into a
//                 sum buffer; after the reduce the sum is scaled by 1/iterations.  The summation
//                 order differs from the sequential running mean: equal to rounding, not bit for bit.
//
// Header-only; link with -lptmi -lrccl.
#pragma once

#include "pathtracer.hpp"
#include "shard_plan.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <mutex>
#include <chrono>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <vector>

struct MultiGpuResult {
    bool ok = false;
    std::string error;
    double seconds = 0.0;          // render + reduce, all devices
    uint64_t rays = 0;             // Intersect calls, all devices
    std::vector<float> image;      // width*height*3, the finished mean image (from device 0)
};

// Renders `iterations` iterations of the scene on devices 0..world-1 and returns the mean image.
// batch: iterations per launch on each device.
//
// A worker runs in two phases.  Phase 1 is everything that can fail on one device alone: stream,
// buffer, scene upload, the render launches and a synchronisation that surfaces their errors.  The
// workers then meet (ShardRendezvous): if any of them failed, none enters the collective and the
// first error is returned.  Phase 2 is the one ncclReduce; should it fail on a device after the
// others have entered it, every communicator is aborted (ncclCommAbort) so that the peers'
// hipStreamSynchronize returns instead of waiting for the missing rank.
// inject_failure_rank: that device reports a failure at the end of phase 1 (tests of the error path).
inline MultiGpuResult RenderMultiGpu(const Camera &camera, int bounces, uint64_t seed, const std::vector<Material> &mv,
                                     const std::vector<Primitive> &pv, const std::vector<Light> &lv, int accel,
                                     uint32_t world, SplitMode split, uint32_t iterations, uint32_t batch,
                                     int inject_failure_rank = -1, const std::vector<Triangle> &tv = std::vector<Triangle>())
{
    MultiGpuResult res;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < (int)world) {
        res.error = "asked for " + std::to_string(world) + " GPUs, " + std::to_string(n_dev) + " visible";
        return res;
    }
    const size_t n = (size_t)camera.Width() * camera.Height() * 3;
    std::vector<int> devs(world);
    for (uint32_t d = 0; d < world; ++d) devs[d] = (int)d;
    std::vector<ncclComm_t> comms(world);
    const ncclResult_t nr = ncclCommInitAll(comms.data(), (int)world, devs.data());
    if (nr != ncclSuccess) { res.error = std::string("ncclCommInitAll: ") + ncclGetErrorString(nr); return res; }

    std::vector<std::string> errors(world);
    std::vector<uint64_t> rays(world, 0);
    std::vector<float *> buffers(world, nullptr);
    std::vector<hipStream_t> streams(world, nullptr);
    ShardRendezvous rendezvous(world);
    std::atomic<bool> aborted{false};
    std::mutex abort_mutex;
    // a collective that cannot complete: abort every communicator, once, so that no peer stays blocked in it
    auto abort_all = [&]() {
        std::lock_guard<std::mutex> lock(abort_mutex);
        if (aborted.exchange(true)) return;
        for (uint32_t r = 0; r < world; ++r) (void)ncclCommAbort(comms[r]);
    };
    res.image.assign(n, 0.f);
    const auto t0 = std::chrono::steady_clock::now();

    auto worker = [&](uint32_t rank) {
        std::string &err = errors[rank];
        Camera cam = camera;
        std::unique_ptr<Pathtracer> pt;
        // ---- phase 1: set-up and render, this device alone ----
        try {
            if (hipSetDevice((int)rank) != hipSuccess) err = "hipSetDevice failed";
            else if (hipStreamCreate(&streams[rank]) != hipSuccess) err = "hipStreamCreate failed";
            else if (hipMalloc((void **)&buffers[rank], n * sizeof(float)) != hipSuccess) err = "hipMalloc failed";
            // pixels this device does not own must be zero for the sum
            else if (hipMemsetAsync(buffers[rank], 0, n * sizeof(float), streams[rank]) != hipSuccess) err = "hipMemsetAsync failed";
            else {
                pt.reset(new Pathtracer(&cam, bounces, seed));
                if (!pt->Init(mv, pv, tv, lv)) err = pt->LastError();
            }
            if (err.empty()) {
                pt->SetAccel(accel);
                pt->SetStream(streams[rank]);
                const ShardPlan plan = MakeShardPlan(split, rank, world, iterations, camera.Height());
                if (split == SplitMode::Sample) pt->SetAccumMode(PT_ACCUM_SUM);
                else pt->SetRowBands(plan.band_rows, rank, world);
                pt->Reset();
                pt->SetIteration(plan.first_iteration - 1);
                for (uint32_t done = 0; done < plan.n_iterations;) {
                    const uint32_t k = plan.n_iterations - done < batch ? plan.n_iterations - done : batch;
                    pt->RunN(buffers[rank], k, /*sync=*/false);
                    done += k;
                }
                if (hipStreamSynchronize(streams[rank]) != hipSuccess) err = "hipStreamSynchronize after the render failed";
            }
            if (err.empty() && (int)rank == inject_failure_rank) err = "injected failure before the reduce";
        } catch (const std::exception &e) {
            err = e.what();
        }
        // ---- all devices meet: the collective is entered by everyone or by no one ----
        const bool everyone_ok = rendezvous.Arrive(err.empty());
        // ---- phase 2: one collective ends the render: sum of the accumulation buffers into device 0 ----
        if (everyone_ok) {
            const ncclResult_t r = ncclReduce(buffers[rank], buffers[rank], n, ncclFloat, ncclSum, 0, comms[rank], streams[rank]);
            if (r != ncclSuccess) { err = std::string("ncclReduce: ") + ncclGetErrorString(r); abort_all(); }
            else if (hipStreamSynchronize(streams[rank]) != hipSuccess) {
                err = aborted.load() ? "reduce aborted after a failure on another device" : "hipStreamSynchronize after the reduce failed";
                abort_all();
            } else {
                try {
                    rays[rank] = pt->RayCount();
                } catch (const std::exception &e) {
                    err = e.what();
                }
                if (rank == 0 && hipMemcpy(res.image.data(), buffers[0], n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
                    err = "hipMemcpy of the image failed";
            }
        }
        try {
            if (pt) pt->Release();
        } catch (const std::exception &) {
        }
    };
    std::vector<std::thread> threads;
    for (uint32_t r = 0; r < world; ++r) threads.emplace_back(worker, r);
    for (auto &t : threads) t.join();
    res.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    for (uint32_t r = 0; r < world; ++r) {
        (void)hipSetDevice((int)r);
        if (buffers[r]) (void)hipFree(buffers[r]);
        if (streams[r]) (void)hipStreamDestroy(streams[r]);
        if (!aborted.load()) (void)ncclCommDestroy(comms[r]);   // ncclCommAbort has freed an aborted communicator
        if (!errors[r].empty() && res.error.empty()) res.error = "device " + std::to_string(r) + ": " + errors[r];
        res.rays += rays[r];
    }
    (void)hipSetDevice(0);
    if (!res.error.empty()) return res;
    if (split == SplitMode::Sample && iterations > 0) {
        const float scale = 1.f / (float)iterations;
        for (float &v : res.image) v *= scale;
    }
    res.ok = true;
    return res;
}
