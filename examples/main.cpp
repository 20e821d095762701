// Headless counterpart of the reference's viewer (main.cpp): builds a scene -- by default the
// demo scene the way InitScene does (main.cpp:128-166), or a built-in benchmark scene, or a scene
// description file ("TODO: configuration file parsing", main.cpp:126) -- renders it through the
// Pathtracer mirror class and writes the running-mean image.  The GLUT window, PBO interop and
// keyboard handling of main.cpp are not reproduced; the "sps" read-out of its title bar
// (main.cpp:366-379) is printed instead.
//
//   pt_headless [--scene demo|synth10k|mixed|FILE] [--width W] [--height H] [--iterations N]
//               [--depth D] [--seed S] [--batch B] [--accel auto|brute] [--out image.ppm|image.pfm]
//               [--checkpoint FILE] [--resume FILE] [--dump-scene]
//               [--gpus N] [--split tile|sample] [--plan]
//
// --iterations N   render until the iteration counter reaches N (4 samples per pixel each)
// --batch B        iterations per launch (default 16; 1 = one Run() per iteration like Display())
// --checkpoint F   write frame buffer + iteration counter to F when done
// --resume F       start from a checkpoint written with the same size, depth and seed
// --dump-scene     print the scene in the scene-file format and exit (touches no GPU)
// --gpus N         shard the render over devices 0..N-1 of this node: one host thread and one
//                  Pathtracer per device, one RCCL reduce(sum) of the accumulation buffers into
//                  device 0 at the end (host/multi_gpu.hpp).  N = 1 goes through the same path
//                  (a communicator of one) when --split is given.
// --split S        tile: 16-row bands dealt round robin, bit-identical to one GPU (default);
//                  sample: contiguous blocks of iterations, summed and scaled
// --plan           print which rows and iterations each device would render and exit (no GPU)
// --inject-failure R   (tests) device R of a sharded render reports a failure before the closing
//                  reduce: the render must end with that error, no device may wait in the collective
#include "multi_gpu.hpp"
#include "pathtracer.hpp"
#include "scene_file.hpp"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// InitScene, main.cpp:128-166
static void InitScene(SceneDescription *s)
{
    s->has_camera = true;
    s->eye = Point(0, 45, 79.5); s->look_at = Point(0, 35, 0); s->up = Vec(0, 1, 0); s->fov = 60.f;
    std::vector<Material> *mv = &s->materials;
    std::vector<Primitive> *pv = &s->primitives;
    mv->push_back(CreateDiffuseMaterial(Color(1.f, 1.f, 1.f), 0.f));
    mv->push_back(CreateDiffuseMaterial(Color(0.75f, 0.25f, 0.25f), 0.f));
    mv->push_back(CreateDiffuseMaterial(Color(0.25f, 0.25f, 0.75f), 0.f));
    mv->push_back(CreateDiffuseMaterial(Color(0.75f, 0.75f, 0.75f), 0.f));
    mv->push_back(CreateSpecularMaterial(Color(0.999f, 0.999f, 0.999f), 1.f));
    mv->push_back(CreateTransmissiveMaterial(Color(0.999f, 0.999f, 0.999f), 1.5f));
    pv->push_back(Primitive(Point(1e5 + 50, 40, 0), 1e5, 2));
    pv->push_back(Primitive(Point(-1e5 - 50, 40, 0), 1e5, 1));
    pv->push_back(Primitive(Point(0, 40, -1e5 - 80), 1e5, 3));
    pv->push_back(Primitive(Point(0, 40, 1e5 + 80), 1e5, 3));
    pv->push_back(Primitive(Point(0, -1e5, 0), 1e5, 3));
    pv->push_back(Primitive(Point(0, 1e5 + 80, 0), 1e5, 3));
    pv->push_back(Primitive(Point(-25, 16.5, -50), 16.5, 4));
    pv->push_back(Primitive(Point(25, 16.5, -25), 16.5, 5));
    pv->push_back(Primitive(Point(0, 579.6, -40), 500, 1, 0));
    s->lights.push_back(Light(8, Color(12.f, 12.f, 12.f)));
}

static int usage(const char *msg)
{
    fprintf(stderr, "pt_headless: %s\n"
            "usage: pt_headless [--scene demo|synth10k|mixed|FILE] [--width W] [--height H] [--iterations N]\n"
            "                   [--depth D] [--seed S] [--batch B] [--accel auto|brute] [--out image.ppm|image.pfm]\n"
            "                   [--checkpoint FILE] [--resume FILE] [--dump-scene]\n"
            "                   [--gpus N] [--split tile|sample] [--plan]\n", msg);
    return 2;
}

static bool ends_with(const std::string &s, const char *suffix)
{
    const size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

int main(int argc, char **argv)
{
    uint32_t width = 640, height = 480;   // main.cpp:388-389
    uint32_t iterations = 64, batch = 16;
    int bounces = 10;                     // main.cpp:177
    uint64_t seed = 1;
    int accel = PT_ACCEL_AUTO;
    bool dump = false, plan_only = false, split_given = false;
    uint32_t gpus = 1;
    int inject_failure = -1;
    SplitMode split = SplitMode::Tile;
    std::string scene = "demo", out = "pt_headless.ppm", checkpoint, resume;

    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--dump-scene") { dump = true; continue; }
        if (a == "--plan") { plan_only = true; continue; }
        if (i + 1 >= argc) return usage(("missing value after " + a).c_str());
        const char *v = argv[++i];
        char *end = nullptr;
        auto number = [&](unsigned long long lo, unsigned long long hi, unsigned long long *dst) {
            const unsigned long long x = strtoull(v, &end, 10);
            if (*v == '\0' || *v == '-' || *end != '\0' || x < lo || x > hi) return false;
            *dst = x;
            return true;
        };
        unsigned long long x = 0;
        if (a == "--scene") scene = v;
        else if (a == "--out") out = v;
        else if (a == "--checkpoint") checkpoint = v;
        else if (a == "--resume") resume = v;
        else if (a == "--accel") {
            if (!strcmp(v, "auto")) accel = PT_ACCEL_AUTO;
            else if (!strcmp(v, "brute")) accel = PT_ACCEL_BRUTE_FORCE;
            else return usage("--accel takes auto or brute");
        }
        else if (a == "--split") {
            if (!strcmp(v, "tile")) split = SplitMode::Tile;
            else if (!strcmp(v, "sample")) split = SplitMode::Sample;
            else return usage("--split takes tile or sample");
            split_given = true;
        }
        else if (a == "--gpus") { if (!number(1, 64, &x)) return usage("bad --gpus"); gpus = (uint32_t)x; }
        else if (a == "--inject-failure") { if (!number(0, 63, &x)) return usage("bad --inject-failure"); inject_failure = (int)x; }
        else if (a == "--width") { if (!number(1, 65536, &x)) return usage("bad --width"); width = (uint32_t)x; }
        else if (a == "--height") { if (!number(1, 65536, &x)) return usage("bad --height"); height = (uint32_t)x; }
        else if (a == "--iterations") { if (!number(0, 1u << 30, &x)) return usage("bad --iterations"); iterations = (uint32_t)x; }
        else if (a == "--batch") { if (!number(1, 1u << 20, &x)) return usage("bad --batch"); batch = (uint32_t)x; }
        else if (a == "--depth") { if (!number(0, 1024, &x)) return usage("bad --depth"); bounces = (int)x; }
        else if (a == "--seed") { if (!number(0, ~0ull, &x)) return usage("bad --seed"); seed = x; }
        else return usage(("unknown option " + a).c_str());
    }
    if (!ends_with(out, ".ppm") && !ends_with(out, ".pfm")) return usage("--out must end in .ppm or .pfm");

    SceneDescription sd;
    if (scene == "demo") InitScene(&sd);
    else if (scene == "synth10k") { if (!BuiltinScene(PT_SCENE_SYNTH10K, 1, &sd)) return 1; }
    else if (scene == "mixed") { if (!BuiltinScene(PT_SCENE_MIXED, 1, &sd)) return 1; }
    else {
        const std::string e = LoadSceneFile(scene, &sd);
        if (!e.empty()) { fprintf(stderr, "pt_headless: %s\n", e.c_str()); return 1; }
        if (!sd.has_camera) { fprintf(stderr, "pt_headless: %s has no camera statement\n", scene.c_str()); return 1; }
    }
    if (dump) { WriteScene(stdout, sd); return 0; }
    if (plan_only) {
        // one line per device: rank, first iteration, iterations, then the row ranges
        for (uint32_t r = 0; r < gpus; ++r) {
            const ShardPlan p = MakeShardPlan(split, r, gpus, iterations, height);
            printf("rank %u first_iteration %u iterations %u rows", r, p.first_iteration, p.n_iterations);
            for (const auto &yy : p.rows) printf(" %u-%u", yy.first, yy.second);
            printf("\n");
        }
        return 0;
    }
    const bool multi = gpus > 1 || split_given;
    if (multi && (!checkpoint.empty() || !resume.empty())) return usage("--checkpoint/--resume work with one GPU only");

    if (multi) {
        try {
            Camera camera(sd.eye, sd.look_at, sd.up, width, height, 0.f, 0.f, sd.fov);
            const MultiGpuResult r = RenderMultiGpu(camera, bounces, seed, sd.materials, sd.primitives, sd.lights, accel,
                                                    gpus, split, iterations, batch, inject_failure, sd.triangles);
            if (!r.ok) { fprintf(stderr, "ERROR: %s\n", r.error.c_str()); return 1; }
            printf("Pathtracer (%.4f sps : %u) on %u GPU(s), %s split, %.3f G bounces/s\n", iterations / r.seconds, iterations,
                   gpus, split == SplitMode::Tile ? "tile" : "sample", (double)r.rays / r.seconds / 1e9);
            const bool ok = ends_with(out, ".pfm") ? WritePFM(out, r.image.data(), width, height)
                                                   : WritePPM(out, r.image.data(), width, height);
            if (!ok) { fprintf(stderr, "ERROR: cannot write %s\n", out.c_str()); return 1; }
            printf("wrote %s\n", out.c_str());
        } catch (const std::exception &e) {
            fprintf(stderr, "ERROR: %s\n", e.what());
            return 1;
        }
        return 0;
    }

    try {
        Camera camera(sd.eye, sd.look_at, sd.up, width, height, 0.f, 0.f, sd.fov);
        Pathtracer pathtracer(&camera, bounces, seed);
        if (!pathtracer.Init(sd.materials, sd.primitives, sd.triangles, sd.lights)) {
            fprintf(stderr, "ERROR: Could not initialize Pathtracer: %s\n", pathtracer.LastError());
            return 1;
        }
        pathtracer.SetAccel(accel);
        float *devBuffer = nullptr;
        const size_t n = (size_t)width * height * 3;
        if (hipMalloc((void **)&devBuffer, n * sizeof(float)) != hipSuccess) { fprintf(stderr, "ERROR: hipMalloc failed\n"); return 1; }
        (void)camera.IsUpdated();   // a fresh camera needs no Reset
        if (!resume.empty()) {
            pathtracer.LoadCheckpoint(devBuffer, resume);
            printf("resumed %s at iteration %u\n", resume.c_str(), pathtracer.GetIteration());
        }

        const uint32_t start = pathtracer.GetIteration();
        const auto t0 = std::chrono::steady_clock::now();
        while (pathtracer.GetIteration() < iterations) {
            if (camera.IsUpdated()) pathtracer.Reset();   // Display(), main.cpp:209
            const uint32_t left = iterations - pathtracer.GetIteration();
            if (batch == 1) pathtracer.Run(devBuffer);
            else pathtracer.RunN(devBuffer, left < batch ? left : batch);
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const uint32_t done = pathtracer.GetIteration() - start;
        if (done) printf("Pathtracer (%.4f sps : %u)\n", done / dt, pathtracer.GetIteration());

        if (!checkpoint.empty()) {
            pathtracer.SaveCheckpoint(devBuffer, checkpoint);
            printf("wrote %s\n", checkpoint.c_str());
        }
        std::vector<float> img(n);
        if (pt_read_image(pathtracer.Handle(), devBuffer, img.data()) != PT_OK) { fprintf(stderr, "ERROR: %s\n", pathtracer.LastError()); return 1; }
        (void)hipFree(devBuffer);
        const bool ok = ends_with(out, ".pfm") ? WritePFM(out, img.data(), width, height) : WritePPM(out, img.data(), width, height);
        if (!ok) { fprintf(stderr, "ERROR: cannot write %s\n", out.c_str()); return 1; }
        printf("wrote %s\n", out.c_str());
        pathtracer.Release();
    } catch (const std::exception &e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    return 0;
}
