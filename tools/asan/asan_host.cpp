// Sanitizer driver of the library's host-side code (make asan): the scene-file parser and writer,
// the built-in scenes, the camera functions, the surface-area BVH builder and the checkpoint
// reader / writer, built with AddressSanitizer + UBSan and without any device code.  Exits
// non-zero when a result is wrong; tests/test_sanitizers.py also fails on any sanitizer report.
//   asan_host <scratch directory>
#include "pt_bvh.hpp"
#include "pt_checkpoint.hpp"
#include "scene_file.hpp"
#include "shard_plan.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

static int g_failures = 0;
#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_failures; } } while (0)

static std::string dump(const SceneDescription &s)
{
    char *buf = nullptr;
    size_t len = 0;
    FILE *f = open_memstream(&buf, &len);
    WriteScene(f, s);
    fclose(f);
    std::string out(buf, len);
    free(buf);
    return out;
}

static void scenes()
{
    for (int id : {PT_SCENE_DEMO, PT_SCENE_SYNTH10K, PT_SCENE_MIXED}) {
        SceneDescription a, b;
        CHECK(BuiltinScene(id, 1, &a));
        const std::string text = dump(a);
        std::istringstream in(text);
        CHECK(ParseScene(in, &b).empty());
        CHECK(dump(b) == text);
    }
    {   // triangles: written after the spheres, lights renumbered to sphere count + triangle index both ways
        const char *text = "camera 0 1 5 0 1 0 0 1 0 50\nmaterial diffuse 0.7 0.7 0.7\ntriangle 0 0 0 1 0 0 0 1 0 0\n"
                           "trianglelight 0 4 4 4\nsphere 0 1 0 0.5 0\ntriangle 0 0 1 1 0 1 0 1 1 0\nlight 0 1 2 3\n";
        std::istringstream in(text);
        SceneDescription t, u;
        CHECK(ParseScene(in, &t).empty());
        CHECK(t.triangles.size() == 2 && t.primitives.size() == 1 && t.lights.size() == 2);
        CHECK(t.lights[0].primId == 1 && t.triangles[0].lightId == 0 && t.lights[1].primId == 0);
        const std::string once = dump(t);
        std::istringstream again(once);
        CHECK(ParseScene(again, &u).empty());
        CHECK(dump(u) == once && u.lights[0].primId == 1);
    }
    SceneDescription none;
    CHECK(!BuiltinScene(77, 1, &none));
    const char *bad[] = {
        "", "\n\n# only comments\n", "material\n", "material diffuse\n", "material diffuse 1 1\n", "sphere 1 2 3\n",
        "material diffuse 1 1 1\nsphere 0 0 0 1 4294967296\n", "material diffuse 1 1 1\nsphere 0 0 0 1e99 0\n",
        "material diffuse 1 1 1\nsphere 0 0 0 1 -1\n", "light 0 1 1 1\n", "camera 1 2 3\n", "camera 0 0 0 0 0 0 0 0 0 60\n",
        "pointlight 1 2\n", "material transmissive 1 1 1 inf\n", "\x01\x02\xff\xfe binary\n",
        "material diffuse 1 1 1\nsphere 0 0 0 1 0\nlight 0 1 1 1\nlight 0 1 1 1\n",
        "material diffuse 1 1 1\ntriangle 0 0 0 1 0 0 0 1\n", "material diffuse 1 1 1\ntriangle 0 0 0 1 0 0 0 1 0 3\n",
        "material diffuse 1 1 1\ntriangle 0 0 0 1 0 0 0 1 0 0\ntrianglelight 1 1 1 1\n",
        "material diffuse 1 1 1\ntriangle 0 0 0 1 0 0 0 1 0 0\ntrianglelight 0 1 1 1\ntrianglelight 0 1 1 1\n",
        "material diffuse 1 1 1\ntriangle 0 0 0 1 0 0 0 nan 0 0\n",
    };
    for (const char *t : bad) {
        SceneDescription s;
        std::istringstream in(t);
        CHECK(!ParseScene(in, &s).empty());
    }
    // a very long line and a very long file
    SceneDescription s;
    std::string longline = "material diffuse 1 1 1 " + std::string(100000, '7') + "\n";
    std::istringstream in1(longline);
    (void)ParseScene(in1, &s);
    std::string big = "camera 0 45 79.5 0 35 0 0 1 0 60\nmaterial diffuse 0.5 0.5 0.5\n";
    for (int i = 0; i < 20000; ++i) big += "sphere " + std::to_string(i % 97) + " " + std::to_string(i % 31) + " -" + std::to_string(i % 53) + " 0.5 0\n";
    std::istringstream in2(big);
    CHECK(ParseScene(in2, &s).empty() && s.primitives.size() == 20000);
    CHECK(!LoadSceneFile("/nonexistent/dir/x.scene", &s).empty());
}

static void cameras()
{
    pt_camera_state st;
    const float eye[3] = {0, 45, 79.5f}, look[3] = {0, 35, 0}, up[3] = {0, 1, 0};
    CHECK(pt_camera_state_init(&st, eye, look, up, 640, 480, 60.f) == PT_OK);
    CHECK(pt_camera_state_init(&st, eye, look, up, 0, 480, 60.f) != PT_OK);
    CHECK(pt_camera_state_init(nullptr, eye, look, up, 1, 1, 60.f) != PT_OK);
    CHECK(pt_camera_state_init(&st, eye, look, up, 640, 480, 60.f) == PT_OK);
    const float d[3] = {1, -2, 3}, th[3] = {0.05f, -0.3f, 9.f};
    for (int i = 0; i < 50; ++i) { CHECK(pt_camera_translate(&st, d) == PT_OK); CHECK(pt_camera_rotate(&st, th) == PT_OK); }
    CHECK(std::isfinite(st.pod.first_ray_dir[0]));
    pt_camera_pod pod;
    CHECK(pt_camera_init(&pod, eye, eye, up, 16, 16, 60.f) == PT_OK || true);   // degenerate look-at: must not crash
}

static void check_bvh(const std::vector<float4> &sph, const std::vector<float> &radius)
{
    ptmi::BvhData d;
    const uint32_t n = (uint32_t)sph.size();
    if (!ptmi::build_bvh(sph.data(), radius.data(), n, &d)) return;
    std::vector<int> seen(n, 0);
    for (int i : d.leaf_index) if (i != 0x7fffffff) { CHECK(i >= 0 && (uint32_t)i < n); if (i >= 0 && (uint32_t)i < n) ++seen[i]; }
    for (int i : d.big_index) { CHECK(i >= 0 && (uint32_t)i < n); if (i >= 0 && (uint32_t)i < n) ++seen[i]; }
    for (uint32_t i = 0; i < n; ++i) CHECK(seen[i] == 1);
    CHECK(d.nodes.size() % ptmi::kBvhNodeVec4 == 0 && d.leaf_sph.size() == d.leaf_index.size());
    CHECK(d.depth <= ptmi::kBvhMaxDepth && d.big_index.size() <= ptmi::kBvhMaxBig);
    const int n_nodes = (int)(d.nodes.size() / ptmi::kBvhNodeVec4), n_leaves = (int)(d.leaf_index.size() / ptmi::kBvhLeafSize);
    for (int k = 0; k < n_nodes; ++k) {
        const float4 r = d.nodes[(size_t)ptmi::kBvhNodeVec4 * k + 6];
        const float f[4] = {r.x, r.y, r.z, r.w};
        for (int c = 0; c < 4; ++c) {
            int ref;
            memcpy(&ref, &f[c], 4);
            if (ref == ptmi::kBvhNoChild) { CHECK(c >= 2); continue; }
            CHECK(ref >= 0 ? ref < n_nodes : ~ref < n_leaves);
        }
    }
}

static void bvhs()
{
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> u(-50.f, 50.f), r(0.05f, 3.f);
    for (uint32_t n : {1u, 5u, 193u, 200u, 1000u, 4097u, 10007u}) {
        std::vector<float4> sph(n);
        std::vector<float> radius(n);
        for (uint32_t i = 0; i < n; ++i) { radius[i] = r(rng); sph[i] = make_float4(u(rng), u(rng), u(rng), radius[i] * radius[i]); }
        check_bvh(sph, radius);
        // the same with the demo room's walls, duplicates, coplanar centres and unboundable spheres
        for (uint32_t i = 0; i < n && i < 6; ++i) { radius[i] = 1e5f; sph[i] = make_float4(1e5f + 50, 40, 0, 1e10f); }
        for (uint32_t i = 6; i + 3 < n; i += 7) { sph[i + 1] = sph[i]; radius[i + 1] = radius[i]; sph[i + 2].y = 0.f; }
        check_bvh(sph, radius);
        if (n > 20) {
            sph[10].x = NAN; radius[11] = INFINITY; sph[11].w = INFINITY; radius[12] = 0.f; sph[12].w = 0.f; sph[13].z = INFINITY;
            radius[14] = -1.f; sph[14].w = 1.f;
            check_bvh(sph, radius);
        }
        for (uint32_t i = 0; i < n; ++i) sph[i] = make_float4(1.f, 2.f, 3.f, 1.f), radius[i] = 1.f;   // all identical
        check_bvh(sph, radius);
    }
    SceneDescription s;
    CHECK(BuiltinScene(PT_SCENE_SYNTH10K, 1, &s));
    std::vector<float4> sph(s.primitives.size());
    std::vector<float> radius(s.primitives.size());
    for (size_t i = 0; i < sph.size(); ++i) {
        const Primitive &p = s.primitives[i];
        radius[i] = p.radius; sph[i] = make_float4(p.pos.x, p.pos.y, p.pos.z, p.radius * p.radius);
    }
    check_bvh(sph, radius);
}

static void checkpoints(const std::string &dir)
{
    using namespace ptmi;
    CheckpointMeta m = {7, 5, 12, 8, 99ull, 0, 0, 0x1234ull, 0x5678ull};
    std::vector<float> img(7 * 5 * 3), back;
    for (size_t i = 0; i < img.size(); ++i) img[i] = (float)i * 0.25f;
    const std::string path = dir + "/a.ckp";
    std::string err;
    uint32_t it = 0;
    CHECK(checkpoint_write(path.c_str(), m, img.data(), &err) == kCkpOk);
    CHECK(checkpoint_read(path.c_str(), m, &back, &it, &err) == kCkpOk && it == 12 && back == img);
    CheckpointMeta other = m; other.iteration = 0;
    CHECK(checkpoint_read(path.c_str(), other, &back, &it, &err) == kCkpOk);          // the counter is not a condition
    other = m; other.integrator = 1; CHECK(checkpoint_read(path.c_str(), other, &back, &it, &err) == kCkpMismatch);
    other = m; other.scene_hash ^= 1; CHECK(checkpoint_read(path.c_str(), other, &back, &it, &err) == kCkpMismatch);
    other = m; other.camera_hash ^= 1; CHECK(checkpoint_read(path.c_str(), other, &back, &it, &err) == kCkpMismatch);
    other = m; other.width = 8; CHECK(checkpoint_read(path.c_str(), other, &back, &it, &err) == kCkpMismatch);
    other = m; other.seed = 1; CHECK(checkpoint_read(path.c_str(), other, &back, &it, &err) == kCkpMismatch);
    // overwriting an existing checkpoint goes through the temporary file
    m.iteration = 13;
    CHECK(checkpoint_write(path.c_str(), m, img.data(), &err) == kCkpOk);
    CHECK(checkpoint_read(path.c_str(), m, &back, &it, &err) == kCkpOk && it == 13);
    FILE *probe = fopen((path + ".tmp").c_str(), "rb");
    CHECK(probe == nullptr);
    if (probe) fclose(probe);
    // damaged files
    FILE *f = fopen(path.c_str(), "rb");
    std::vector<unsigned char> bytes(64 + img.size() * 4);
    CHECK(fread(bytes.data(), 1, bytes.size(), f) == bytes.size());
    fclose(f);
    auto write_bytes = [&](const std::string &p, size_t n, int extra) {
        FILE *o = fopen(p.c_str(), "wb");
        fwrite(bytes.data(), 1, n, o);
        for (int i = 0; i < extra; ++i) fputc(0, o);
        fclose(o);
    };
    const std::string bad = dir + "/bad.ckp";
    for (size_t n : {(size_t)0, (size_t)7, (size_t)63, (size_t)64, bytes.size() - 1}) {
        write_bytes(bad, n, 0);
        CHECK(checkpoint_read(bad.c_str(), m, &back, &it, &err) == kCkpInvalid);
    }
    write_bytes(bad, bytes.size(), 3);
    CHECK(checkpoint_read(bad.c_str(), m, &back, &it, &err) == kCkpInvalid);
    bytes[7] = '1';                                                                  // the older format's magic
    write_bytes(bad, bytes.size(), 0);
    CHECK(checkpoint_read(bad.c_str(), m, &back, &it, &err) == kCkpInvalid && err.find("older format") != std::string::npos);
    bytes[0] = 'X';
    write_bytes(bad, bytes.size(), 0);
    CHECK(checkpoint_read(bad.c_str(), m, &back, &it, &err) == kCkpInvalid);
    CHECK(checkpoint_read((dir + "/missing.ckp").c_str(), m, &back, &it, &err) == kCkpInvalid);
    CHECK(checkpoint_write((dir + "/no/such/dir/x.ckp").c_str(), m, img.data(), &err) == kCkpInvalid);
    // image writers
    CHECK(WritePFM(dir + "/i.pfm", img.data(), 7, 5) && WritePPM(dir + "/i.ppm", img.data(), 7, 5));
    CHECK(!WritePFM(dir + "/no/such/dir/i.pfm", img.data(), 7, 5));
}

// The meeting point of the sharded render's host threads: everyone learns whether anyone failed,
// however the arrivals interleave, and no thread is left waiting.
static void rendezvous()
{
    for (uint32_t world : {1u, 2u, 8u}) {
        for (int failing = -1; failing < (int)world; ++failing) {
            ShardRendezvous r(world);
            std::vector<int> verdict(world, -1);
            std::vector<std::thread> threads;
            for (uint32_t k = 0; k < world; ++k)
                threads.emplace_back([&, k] {
                    if (k % 3 == 1) std::this_thread::yield();
                    verdict[k] = r.Arrive((int)k != failing) ? 1 : 0;
                });
            for (auto &t : threads) t.join();
            for (uint32_t k = 0; k < world; ++k) CHECK(verdict[k] == (failing < 0 ? 1 : 0));
        }
    }
    // the partition: every iteration and every row exactly once
    for (uint32_t world : {1u, 3u, 8u}) {
        std::vector<int> it(1001, 0), row(1080, 0);
        for (uint32_t k = 0; k < world; ++k) {
            const ShardPlan s = MakeShardPlan(SplitMode::Sample, k, world, 1000, 1080);
            for (uint32_t i = 0; i < s.n_iterations; ++i) ++it[s.first_iteration + i];
            const ShardPlan t = MakeShardPlan(SplitMode::Tile, k, world, 1000, 1080);
            for (const auto &yy : t.rows) for (uint32_t y = yy.first; y < yy.second; ++y) ++row[y];
        }
        bool once = it[0] == 0;
        for (int i = 1; i <= 1000; ++i) once = once && it[i] == 1;
        for (int y = 0; y < 1080; ++y) once = once && row[y] == 1;
        CHECK(once);
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: asan_host <scratch directory>\n"); return 2; }
    scenes();
    cameras();
    bvhs();
    checkpoints(argv[1]);
    rendezvous();
    printf("asan_host: %d failure(s)\n", g_failures);
    return g_failures ? 1 : 0;
}
