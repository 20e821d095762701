/* Sanitizer driver of the CPU oracle (make asan): renders small cases of every kind the tests use --
 * both integrators, several threads, a sub-rectangle with per-path outputs, an empty scene, depth 0,
 * a NaN coordinate, unknown light and material types -- in a build with AddressSanitizer and UBSan,
 * and prints an FNV-1a hash per case, which tests/test_sanitizers.py compares with the plain
 * build's.  Test infrastructure only. */
#include "pt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static unsigned long long fnv(const void *p, size_t n, unsigned long long h)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

static int run_case(const char *name, int integrator, const ptor_material *m, uint32_t nm, const ptor_sphere *s, uint32_t ns,
                    const ptor_light *l, uint32_t nl, uint32_t w, uint32_t h, int depth, uint32_t first, uint32_t n_it,
                    uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, int threads)
{
    const float eye[3] = {0, 45, 79.5f}, look[3] = {0, 35, 0}, up[3] = {0, 1, 0};
    ptor_camera cam;
    ptor_camera_init(&cam, eye, look, up, w, h, 60.f);
    const size_t n_paths = (size_t)n_it * (y1 - y0) * (x1 - x0) * 4;
    float *img = (float *)calloc((size_t)w * h * 3, sizeof(float));
    int32_t *hits = (int32_t *)malloc(n_paths * (size_t)(depth + 1) * sizeof(int32_t));
    float *rad = (float *)malloc(n_paths * 3 * sizeof(float));
    uint64_t rays = 0;
    const int rc = ptor_render_with(integrator, &cam, m, nm, s, ns, l, nl, 7, depth, first, n_it, x0, y0, x1, y1, 0, img,
                                    hits, rad, &rays, threads);
    unsigned long long hsh = 1469598103934665603ull;
    hsh = fnv(img, (size_t)w * h * 3 * sizeof(float), hsh);
    hsh = fnv(hits, n_paths * (size_t)(depth + 1) * sizeof(int32_t), hsh);
    hsh = fnv(rad, n_paths * 3 * sizeof(float), hsh);
    printf("%s rc=%d rays=%llu hash=%016llx\n", name, rc, (unsigned long long)rays, hsh);
    free(img); free(hits); free(rad);
    return rc;
}

int main(void)
{
    /* the demo scene's data (main.cpp:133-164 of the reference) */
    ptor_material m[6] = {{{1, 1, 1}, 0, 0}, {{.75f, .25f, .25f}, 0, 0}, {{.25f, .25f, .75f}, 0, 0}, {{.75f, .75f, .75f}, 0, 0},
                          {{.999f, .999f, .999f}, 1, 1}, {{.999f, .999f, .999f}, 1.5f, 2}};
    ptor_sphere s[10] = {{{1e5f + 50, 40, 0}, 1e5f, 2, -1}, {{-1e5f - 50, 40, 0}, 1e5f, 1, -1}, {{0, 40, -1e5f - 80}, 1e5f, 3, -1},
                         {{0, 40, 1e5f + 80}, 1e5f, 3, -1}, {{0, -1e5f, 0}, 1e5f, 3, -1}, {{0, 1e5f + 80, 0}, 1e5f, 3, -1},
                         {{-25, 16.5f, -50}, 16.5f, 4, -1}, {{25, 16.5f, -25}, 16.5f, 5, -1}, {{0, 579.6f, -40}, 500, 1, 0},
                         {{5, 30, -20}, 4, 0, 1}};
    ptor_light l[3];
    memset(l, 0, sizeof l);
    l[0].type = 1; l[0].u.prim_id = 8; l[0].intensity[0] = l[0].intensity[1] = l[0].intensity[2] = 12;
    l[1].type = 0; l[1].u.pos[0] = 10; l[1].u.pos[1] = 60; l[1].u.pos[2] = -30; l[1].intensity[0] = 400; l[1].intensity[1] = 300; l[1].intensity[2] = 200;
    l[2].type = 5; l[2].u.prim_id = 0xdeadbeefu; l[2].intensity[0] = 9;   /* unknown type: samples to black */
    int bad = 0;
    bad |= run_case("demo_path", 0, m, 6, s, 9, l, 1, 40, 24, 8, 1, 2, 0, 0, 40, 24, 1);
    bad |= run_case("demo_path_threads", 0, m, 6, s, 9, l, 1, 40, 24, 8, 3, 2, 0, 0, 40, 24, 5);
    bad |= run_case("demo_rect", 0, m, 6, s, 9, l, 1, 40, 24, 16, 1, 1, 7, 3, 29, 20, 3);
    bad |= run_case("demo_nee", 1, m, 6, s, 9, l, 1, 40, 24, 8, 1, 2, 0, 0, 40, 24, 4);
    bad |= run_case("three_lights_nee", 1, m, 6, s, 10, l, 3, 33, 17, 6, 1, 2, 0, 0, 33, 17, 2);
    bad |= run_case("depth0", 0, m, 6, s, 9, l, 1, 17, 9, 0, 1, 3, 0, 0, 17, 9, 2);
    bad |= run_case("empty", 0, m, 1, s, 0, l, 0, 16, 16, 4, 1, 1, 0, 0, 16, 16, 2);
    bad |= run_case("one_pixel", 1, m, 6, s, 9, l, 1, 1, 1, 8, 1, 9, 0, 0, 1, 1, 3);
    s[7].pos[1] = NAN; s[6].radius = 0.f; m[3].type = 9;   /* NaN coordinate, radius 0, unknown material type */
    bad |= run_case("weird", 0, m, 6, s, 9, l, 1, 24, 16, 8, 1, 2, 0, 0, 24, 16, 2);
    bad |= run_case("weird_nee", 1, m, 6, s, 9, l, 1, 24, 16, 8, 1, 2, 0, 0, 24, 16, 2);
    return bad ? 1 : 0;
}
