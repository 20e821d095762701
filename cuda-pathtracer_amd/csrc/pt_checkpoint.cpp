// Checkpoint files (see pt_checkpoint.hpp).  Plain host code: no device calls, so that it also
// builds and runs under AddressSanitizer / UBSan on a machine without a GPU (make asan).
#include "pt_checkpoint.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace ptmi {

namespace {

struct FileHeader {
    char magic[8];        // "PTMICKP2"
    uint32_t width, height;
    uint32_t iteration;
    int32_t max_bounces;
    uint64_t seed;
    int32_t accum_mode;
    int32_t integrator;
    uint64_t scene_hash;
    uint64_t camera_hash;
    uint32_t reserved[2];
};
static_assert(sizeof(FileHeader) == 64, "checkpoint header layout");
const char kMagic[8] = {'P', 'T', 'M', 'I', 'C', 'K', 'P', '2'};
const char kMagicV1[8] = {'P', 'T', 'M', 'I', 'C', 'K', 'P', '1'};

std::string format(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
std::string format(const char *fmt, ...)
{
    char buf[640];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return buf;
}

} // namespace

uint64_t fingerprint(const void *data, size_t bytes, uint64_t h)
{
    const unsigned char *p = static_cast<const unsigned char *>(data);
    for (size_t i = 0; i < bytes; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}

int checkpoint_write(const char *path, const CheckpointMeta &m, const float *image, std::string *error)
{
    const size_t n = (size_t)m.width * m.height * 3;
    FileHeader hd;
    memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, kMagic, 8);
    hd.width = m.width; hd.height = m.height;
    hd.iteration = m.iteration;
    hd.max_bounces = m.max_bounces;
    hd.seed = m.seed;
    hd.accum_mode = m.accum_mode;
    hd.integrator = m.integrator;
    hd.scene_hash = m.scene_hash;
    hd.camera_hash = m.camera_hash;
    const std::string tmp = std::string(path) + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) { *error = format("cannot open %s for writing", tmp.c_str()); return kCkpInvalid; }
    bool ok = fwrite(&hd, sizeof hd, 1, f) == 1 && fwrite(image, sizeof(float), n, f) == n;
    ok = fflush(f) == 0 && ok;
    ok = fclose(f) == 0 && ok;
    if (!ok) { remove(tmp.c_str()); *error = format("short write to %s", tmp.c_str()); return kCkpInvalid; }
    if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); *error = format("cannot rename %s to %s", tmp.c_str(), path); return kCkpInvalid; }
    return kCkpOk;
}

int checkpoint_read(const char *path, const CheckpointMeta &e, std::vector<float> *image, uint32_t *iteration,
                    std::string *error)
{
    FILE *f = fopen(path, "rb");
    if (!f) { *error = format("cannot open %s", path); return kCkpInvalid; }
    FileHeader hd;
    if (fread(&hd, sizeof hd, 1, f) != 1 || (memcmp(hd.magic, kMagic, 8) != 0 && memcmp(hd.magic, kMagicV1, 8) != 0)) {
        fclose(f);
        *error = format("%s is not a checkpoint", path);
        return kCkpInvalid;
    }
    if (memcmp(hd.magic, kMagicV1, 8) == 0) {
        fclose(f);
        *error = format("%s is a checkpoint of an older format (no integrator / scene / camera fingerprints); render it again", path);
        return kCkpInvalid;
    }
    std::string diff;
    if (hd.width != e.width || hd.height != e.height) diff = format("size %ux%u vs %ux%u", hd.width, hd.height, e.width, e.height);
    else if (hd.max_bounces != e.max_bounces) diff = format("depth %d vs %d", hd.max_bounces, e.max_bounces);
    else if (hd.seed != e.seed) diff = format("seed %llu vs %llu", (unsigned long long)hd.seed, (unsigned long long)e.seed);
    else if (hd.accum_mode != e.accum_mode) diff = format("accumulation mode %d vs %d", hd.accum_mode, e.accum_mode);
    else if (hd.integrator != e.integrator) diff = format("integrator %d vs %d", hd.integrator, e.integrator);
    else if (hd.scene_hash != e.scene_hash) diff = "another scene (fingerprints of the material / sphere / light arrays differ)";
    else if (hd.camera_hash != e.camera_hash) diff = "another camera";
    if (!diff.empty()) {
        fclose(f);
        *error = format("%s was rendered under other conditions than this instance (file vs instance): %s", path, diff.c_str());
        return kCkpMismatch;
    }
    const size_t n = (size_t)e.width * e.height * 3;
    image->resize(n);
    const bool ok = fread(image->data(), sizeof(float), n, f) == n && fgetc(f) == EOF;
    fclose(f);
    if (!ok) { *error = format("%s is truncated or has trailing bytes", path); return kCkpInvalid; }
    *iteration = hd.iteration;
    return kCkpOk;
}

} // namespace ptmi
