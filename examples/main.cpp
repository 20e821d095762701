// Headless counterpart of the reference's viewer: builds the demo scene the way InitScene does
// (main.cpp:128-166), renders `iterations` frames through the Pathtracer mirror class and
// writes the running-mean image as a binary PPM.  The GLUT window, PBO interop and keyboard
// handling of main.cpp are not reproduced; the "sps" read-out of its title bar
// (main.cpp:366-379) is printed instead.
//
//   pt_headless [width height iterations max_bounces out.ppm]
#include "pathtracer.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

static void InitScene(Camera *camera, uint32_t width, uint32_t height, std::vector<Material> *mv,
                      std::vector<Primitive> *pv, std::vector<Light> *lv)
{
    *camera = Camera(Point(0, 45, 79.5), Point(0, 35, 0), Vec(0, 1, 0), width, height, 0.f, 0.f, 60.f);
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
    lv->push_back(Light(8, Color(12.f, 12.f, 12.f)));
}

int main(int argc, char **argv)
{
    const uint32_t width = argc > 1 ? (uint32_t)atoi(argv[1]) : 640;   // main.cpp:388-389
    const uint32_t height = argc > 2 ? (uint32_t)atoi(argv[2]) : 480;
    const int iterations = argc > 3 ? atoi(argv[3]) : 64;
    const int bounces = argc > 4 ? atoi(argv[4]) : 10;                 // main.cpp:177
    const char *out = argc > 5 ? argv[5] : "pt_headless.ppm";

    Camera camera;
    std::vector<Material> materials;
    std::vector<Primitive> primitives;
    std::vector<Light> lights;
    InitScene(&camera, width, height, &materials, &primitives, &lights);

    try {
        Pathtracer pathtracer(&camera, bounces);
        if (!pathtracer.Init(materials, primitives, lights)) {
            fprintf(stderr, "ERROR: Could not initialize Pathtracer: %s\n", pathtracer.LastError());
            return 1;
        }
        float *devBuffer = nullptr;
        const size_t n = (size_t)width * height * 3;
        if (hipMalloc((void **)&devBuffer, n * sizeof(float)) != hipSuccess) return 1;

        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iterations; ++i) {
            if (camera.IsUpdated()) pathtracer.Reset();   // Display(), main.cpp:209
            pathtracer.Run(devBuffer);
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("Pathtracer (%.4f sps : %u)\n", iterations / dt, pathtracer.GetIteration());

        std::vector<float> img(n);
        if (hipMemcpy(img.data(), devBuffer, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        (void)hipFree(devBuffer);
        FILE *f = fopen(out, "wb");
        if (!f) return 1;
        fprintf(f, "P6\n%u %u\n255\n", width, height);
        for (size_t i = 0; i < n; ++i) {
            const float c = std::min(1.f, std::max(0.f, img[i]));
            fputc((int)(powf(c, 1.f / 2.2f) * 255.f + 0.5f), f);
        }
        fclose(f);
        printf("wrote %s\n", out);
        pathtracer.Release();
    } catch (const std::exception &e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    return 0;
}
