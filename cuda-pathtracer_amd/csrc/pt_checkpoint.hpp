// Checkpoint files of a render (host code only, no device calls): a 64-byte header and
// width*height*3 floats.  With the counter-based random numbers the frame buffer, the iteration
// counter and the seed are the whole state of a render; the header also pins what the frame was
// rendered WITH (size, depth, accumulation mode, integrator, fingerprints of the scene arrays and
// of the camera), so that a resume under different conditions is refused instead of blending two
// different estimators.  The reference has no checkpoint format (its state dies with the window,
// pathtracer.h:58).
#pragma once

#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

namespace ptmi {

struct CheckpointMeta {
    uint32_t width, height;
    uint32_t iteration;
    int32_t max_bounces;
    uint64_t seed;
    int32_t accum_mode;
    int32_t integrator;
    uint64_t scene_hash;    // fingerprint() of the material, sphere and light arrays given to pt_init_scene
    uint64_t camera_hash;   // fingerprint() of the camera pod
};

// FNV-1a over `bytes` bytes, continuing from `h` (start with kFingerprintSeed).
constexpr uint64_t kFingerprintSeed = 1469598103934665603ull;
uint64_t fingerprint(const void *data, size_t bytes, uint64_t h = kFingerprintSeed);

// Return values of the two functions below (they mirror pt_status without including it).
enum { kCkpOk = 0, kCkpInvalid = -1, kCkpMismatch = -2 };

// Writes `path` atomically: the data goes to path + ".tmp", is flushed, and replaces `path` by
// rename, so that a crash mid-write leaves the previous checkpoint intact.
int checkpoint_write(const char *path, const CheckpointMeta &meta, const float *image, std::string *error);

// Reads `path`.  Everything in `expect` except `iteration` must match the file (kCkpMismatch
// otherwise, with the differing field named); a file that is not a checkpoint, is truncated or has
// trailing bytes is kCkpInvalid.  On success `image` holds width*height*3 floats and *iteration
// the stored counter.
int checkpoint_read(const char *path, const CheckpointMeta &expect, std::vector<float> *image, uint32_t *iteration,
                    std::string *error);

} // namespace ptmi
