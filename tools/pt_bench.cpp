// Quick A/B harness over the C ABI (no Python start-up): renders a built-in workload a few
// times and prints time, bounces/s and an FNV-1a hash of the frame buffer, so that a kernel
// change can be checked for speed and for bit-identical output in one short GPU call.
//   pt_bench <scene 0|1|2> <width> <height> <iterations> <depth> <repeats> [variants]
// variants: comma-separated pt_set_kernel_variant values; +1000 selects the brute-force scan
// (PT_ACCEL_BRUTE_FORCE), +2000 the NEE + MIS integrator for that entry.
#include "ptmi.h"
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

// The shorthand codes of the tuning experiments (a variant plus one pt_debug_set_tuning call):
// 2..64 regeneration threshold, 300 / 301 camera-ray ring off / on, 500 + 1 stealing without the
// ring, + 2 / + 3 unit of 1 / 4 paths, + 10 + k: k + 1 iteration ranges, 100 + 1 static wavefront,
// 100 + k refill threshold k, 999 PT_VARIANT_AUTO.
static int set_variant(pt_handle h, int v)
{
    int base = v, key = -1, value = 0;
    if (v == 999) base = PT_VARIANT_AUTO;   // so that -1 can be combined with + 1000 / + 2000: 1999, 2999
    else if (v >= 2 && v <= 64) { base = PT_VARIANT_REGENERATION; key = PT_TUNE_REGEN_MIN_IDLE; value = v; }
    else if (v == 300 || v == 301) { base = PT_VARIANT_REGENERATION; key = PT_TUNE_CAMERA_RAY_RING; value = v - 300; }
    else if (v > PT_VARIANT_WAVEFRONT && v <= PT_VARIANT_WAVEFRONT + 64) { base = PT_VARIANT_WAVEFRONT; key = PT_TUNE_WAVEFRONT_REFILL; value = v - 100 == 1 ? 0 : v - 100; }
    else if (v == PT_VARIANT_STEALING + 1) { base = PT_VARIANT_STEALING; key = PT_TUNE_CAMERA_RAY_RING; value = 0; }
    else if (v == PT_VARIANT_STEALING + 2 || v == PT_VARIANT_STEALING + 3) { base = PT_VARIANT_STEALING; key = PT_TUNE_STEAL_UNIT; value = v == PT_VARIANT_STEALING + 2 ? 1 : 4; }
    else if (v >= PT_VARIANT_STEALING + 10 && v <= PT_VARIANT_STEALING + 26) { base = PT_VARIANT_STEALING; key = PT_TUNE_ITERATION_RANGES; value = v - (PT_VARIANT_STEALING + 10) + 1; }
    int rc = pt_set_kernel_variant(h, base);
    if (rc == PT_OK && key >= 0) rc = pt_debug_set_tuning(h, key, value);
    return rc;
}

int main(int argc, char **argv)
{
    const int scene = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t w = argc > 2 ? atoi(argv[2]) : 1920, h = argc > 3 ? atoi(argv[3]) : 1080;
    const uint32_t iters = argc > 4 ? atoi(argv[4]) : 64;
    const int depth = argc > 5 ? atoi(argv[5]) : 8;
    const int repeats = argc > 6 ? atoi(argv[6]) : 3;
    // comma-separated list of kernel variants, run interleaved in this one process
    std::vector<int> variants;
    {
        const char *v = argc > 7 ? argv[7] : "0";
        for (const char *q = v; *q;) { variants.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q) ++q; }
    }

    uint32_t nm = 0, ns = 0, nl = 0;
    float eye[3], look[3], up[3], fov;
    pt_scene_build(scene, 1, nullptr, &nm, nullptr, &ns, nullptr, &nl, eye, look, up, &fov);
    std::vector<pt_material> m(nm); std::vector<pt_sphere> s(ns); std::vector<pt_light> l(nl);
    pt_scene_build(scene, 1, m.data(), &nm, s.data(), &ns, l.data(), &nl, nullptr, nullptr, nullptr, nullptr);
    pt_camera_pod cam;
    pt_camera_init(&cam, eye, look, up, w, h, fov);
    pt_handle hd;
    if (pt_create(&hd, &cam, depth, 1) != PT_OK) { printf("%s\n", pt_last_error(nullptr)); return 1; }
    if (pt_init_scene(hd, m.data(), nm, s.data(), ns, l.data(), nl) != PT_OK) { printf("%s\n", pt_last_error(hd)); return 1; }
    printf("spheres %u, bvh nodes %u\n", ns, pt_accel_nodes(hd));
    float *buf;
    const size_t n = (size_t)w * h * 3;
    if (hipMalloc((void **)&buf, n * sizeof(float)) != hipSuccess) return 1;
    std::vector<float> img(n);
    for (int r = 0; r < repeats * (int)variants.size(); ++r) {
        const int variant = variants[r % variants.size()];
        pt_set_accel(hd, variant % 2000 >= 1000 ? PT_ACCEL_BRUTE_FORCE : PT_ACCEL_AUTO);
        pt_set_integrator(hd, variant >= 2000 ? PT_INTEGRATOR_NEE_MIS : PT_INTEGRATOR_PATH);
        if (set_variant(hd, variant % 1000) != PT_OK) { printf("%s\n", pt_last_error(hd)); return 1; }
        pt_reset(hd); pt_reset_counters(hd);
        const auto t0 = std::chrono::steady_clock::now();
        if (pt_run_n(hd, buf, iters, 1) != PT_OK) { printf("%s\n", pt_last_error(hd)); return 1; }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        float ms = 0; pt_last_kernel_ms(hd, &ms);
        uint64_t rays = 0; pt_ray_count(hd, &rays);
        pt_read_image(hd, buf, img.data());
        uint64_t hash = 1469598103934665603ull;
        const unsigned char *p = (const unsigned char *)img.data();
        for (size_t i = 0; i < n * 4; ++i) { hash ^= p[i]; hash *= 1099511628211ull; }
        printf("scene %d %ux%u it%u d%d v%d: wall %.3f ms kernel %.3f ms  %.3f Gbounces/s  rays %llu  hash %016llx\n",
               scene, w, h, iters, depth, variant, dt * 1e3, ms, rays / (ms * 1e-3) / 1e9,
               (unsigned long long)rays, (unsigned long long)hash);
        uint64_t ls[64];
        if (pt_lane_stats(hd, ls) == PT_OK && ls[8]) {
            static const char *site[32] = {"ring refill", "path start", "sphere test", "sphere sqrt part", "loop pass",
                "bounce shading", "emission", "throughput update", "russian roulette", "", "bsdf sample", "glass part 1",
                "diffuse part 1", "glass part 2", "diffuse part 2", "mirror direction", "normalised direction",
                "lanes with a ray", "path end", "pixel-iteration end", "steal attempt", "steal", "bvh box refinement", "bvh inner step", "bvh leaf step"};
            printf("  lane statistics: site, wave executions, executions per loop pass, mean active lanes / 64\n");
            for (int i = 0; i < 25; ++i)
                if (ls[2 * i])
                    printf("    %2d %-22s %12llu  %7.3f  %6.3f\n", i, site[i], (unsigned long long)ls[2 * i],
                           (double)ls[2 * i] / (double)ls[8], (double)ls[2 * i + 1] / (64.0 * (double)ls[2 * i]));
        }
        uint64_t st[32];
        if (pt_accel_stats(hd, st) == PT_OK && st[2]) {
            printf("  bvh: %.1f inner nodes + %.1f leaves per ray, %llu rays, %llu over budget, %llu long-direction\n",
                   (double)st[0] / st[2], (double)st[1] / st[2], (unsigned long long)st[2],
                   (unsigned long long)st[3], (unsigned long long)st[7]);
            printf("  per wave traversal (%llu): %.1f inner + %.1f leaf iterations\n  visits histogram (log2):",
                   (unsigned long long)st[6], (double)st[4] / st[6], (double)st[5] / st[6]);
            for (int k = 0; k < 16; ++k) printf(" %llu", (unsigned long long)st[8 + k]);
            printf("\n");
        }
    }
    pt_destroy(hd);
    return 0;
}
