// Host-only pieces of the sharded render (no HIP, no RCCL): which rows and iterations a device
// renders, and the meeting point at which the per-device host threads agree whether the closing
// collective may be entered.  The reference is single-GPU (main.cpp:386-409); this has no
// counterpart there.
#pragma once

#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <utility>
#include <vector>

enum class SplitMode { Tile, Sample };

// What one device renders.
struct ShardPlan {
    uint32_t first_iteration = 1;   // 1-based, globally numbered
    uint32_t n_iterations = 0;
    uint32_t band_rows = 16;        // tile split: bands of this many rows, band b -> device b % world
    std::vector<std::pair<uint32_t, uint32_t>> rows;   // [y0, y1) ranges this device renders
};

// Partition arithmetic.  Sample split: iterations are dealt out in contiguous blocks, the first
// (iterations % world) devices get one more.  Tile split: 16-row bands, round robin.
inline ShardPlan MakeShardPlan(SplitMode split, uint32_t rank, uint32_t world, uint32_t iterations, uint32_t height)
{
    ShardPlan p;
    if (split == SplitMode::Sample) {
        const uint32_t base = iterations / world, extra = iterations % world;
        p.n_iterations = base + (rank < extra ? 1u : 0u);
        p.first_iteration = rank * base + (rank < extra ? rank : extra) + 1;
        p.rows.emplace_back(0u, height);
    } else {
        p.n_iterations = iterations;
        const uint32_t n_bands = (height + p.band_rows - 1) / p.band_rows;
        for (uint32_t b = rank; b < n_bands; b += world)
            p.rows.emplace_back(b * p.band_rows, (b + 1) * p.band_rows < height ? (b + 1) * p.band_rows : height);
    }
    return p;
}

// A collective blocks until every rank has entered it, so a rank that failed before the collective
// would leave its peers waiting forever.  Every worker thread therefore calls Arrive(ok) exactly
// once, after its own set-up and render and before the collective; the call returns when all
// `world` workers have arrived, and returns true only if every one of them arrived with ok == true.
// On false nobody enters the collective.
class ShardRendezvous {
public:
    explicit ShardRendezvous(uint32_t world) : expected_(world) {}
    bool Arrive(bool ok)
    {
        std::unique_lock<std::mutex> lock(m_);
        all_ok_ = all_ok_ && ok;
        if (++arrived_ == expected_) cv_.notify_all();
        else cv_.wait(lock, [this] { return arrived_ == expected_; });
        return all_ok_;
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    const uint32_t expected_;
    uint32_t arrived_ = 0;
    bool all_ok_ = true;
};
