// C++ host-side mirror of the reference's scene and renderer interface over libptmi's C ABI.
//
// A caller written against the reference's headers -- InitScene/InitPathtracer/Display in
// main.cpp:128-232 -- compiles against this header with the same spelling:
//
//     Camera camera(Point(0,45,79.5), Point(0,35,0), Vec(0,1,0), width, height, 0.f, 0.f, 60.f);
//     std::vector<Material> mv;  mv.push_back(CreateDiffuseMaterial(Color(.75f,.25f,.25f), 0.f));
//     std::vector<Primitive> pv; pv.push_back(Primitive(Point(0,-1e5,0), 1e5, 3));
//     std::vector<Light> lv;     lv.push_back(Light(8, Color(12.f,12.f,12.f)));
//     Pathtracer pt(&camera, 10);
//     pt.Init(mv, pv, lv);
//     pt.Run(devBuffer);           // device pointer to width*height*3 floats
//
// Only what the trace loop consumes is mirrored (pathtracer.h:38-65, camera.h:28-72,
// material.h:25-77, primitive.h:28-37, light.h:34-53).  Header-only; link with -lptmi.
// Errors: Init returns false and Run throws std::runtime_error with pt_last_error() text (the
// reference prints and exit(-1)s, cutil.h:27-52).
#pragma once

#include "ptmi.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

struct Vec {
    float x, y, z, w;
    Vec(float x_ = 0.f, float y_ = 0.f, float z_ = 0.f, float w_ = 0.f) : x(x_), y(y_), z(z_), w(w_) {}
};

struct Point {
    float x, y, z;
    Point(float x_ = 0.f, float y_ = 0.f, float z_ = 0.f) : x(x_), y(y_), z(z_) {}
};

struct Color {
    float r, g, b;
    Color(float r_ = 0.f, float g_ = 0.f, float b_ = 0.f) : r(r_), g(g_), b(b_) {}
};

// camera.h:28-72.  The derived image-plane vectors are computed by pt_camera_init, which
// evaluates the constructor's and UpdateParameter()'s expressions in the reference's order.
class Camera {
public:
    Camera() : state_{} {}
    Camera(const Point &eye, const Point &lookAtP, const Vec &up, uint32_t iw, uint32_t ih,
           float /*lensRadius*/, float /*focalDistance*/, float fov)
    {
        const float e[3] = {eye.x, eye.y, eye.z}, l[3] = {lookAtP.x, lookAtP.y, lookAtP.z}, u[3] = {up.x, up.y, up.z};
        if (pt_camera_state_init(&state_, e, l, u, iw, ih, fov) != PT_OK) throw std::invalid_argument("Camera: bad arguments");
        updated_ = true;
    }
    // camera.h:79-90
    void Translate(const Vec &delta)
    {
        const float d[3] = {delta.x, delta.y, delta.z};
        pt_camera_translate(&state_, d);
        updated_ = true;
    }
    // camera.h:97-129
    void Rotate(const Vec &theta)
    {
        const float t[3] = {theta.x, theta.y, theta.z};
        pt_camera_rotate(&state_, t);
        updated_ = true;
    }
    uint32_t Width() const { return state_.pod.width; }
    uint32_t Height() const { return state_.pod.height; }
    Point GetPos() const { return Point(state_.pod.pos[0], state_.pod.pos[1], state_.pod.pos[2]); }
    // camera.h:134
    bool IsUpdated() { const bool b = updated_; updated_ = false; return b; }
    const pt_camera_pod &Pod() const { return state_.pod; }

private:
    pt_camera_state state_;
    bool updated_ = false;
};

enum MaterialType { DIFFUSE = PT_DIFFUSE, SPECULAR = PT_SPECULAR, TRANSMISSIVE = PT_TRANSMISSIVE };

// material.h:31-48
class Material {
public:
    Color color;
    float coef = 0.f;
    MaterialType type = DIFFUSE;
};

inline Material MakeMaterial(const Color &c, float coef, MaterialType t)
{
    Material m;
    m.color = c; m.coef = coef; m.type = t;
    return m;
}
// material.h:55-77
inline Material CreateDiffuseMaterial(const Color &c, float coef) { return MakeMaterial(c, coef, DIFFUSE); }
inline Material CreateSpecularMaterial(const Color &c, float coef) { return MakeMaterial(c, coef, SPECULAR); }
inline Material CreateTransmissiveMaterial(const Color &c, float coef) { return MakeMaterial(c, coef, TRANSMISSIVE); }

// primitive.h:28-37 (a sphere); the bounding box the reference stores is never read by the
// trace loop and is not kept.
class Primitive {
public:
    Primitive(const Point &p, float rad, uint32_t matId) : pos(p), materialId(matId), radius(rad), lightId(-1) {}
    Primitive(const Point &p, float rad, uint32_t matId, int lightId_) : pos(p), materialId(matId), radius(rad), lightId(lightId_) {}
    bool IsLight() const { return lightId > -1; }
    Point pos;
    uint32_t materialId;
    float radius;
    int lightId;
};

// Not in the reference (its only primitive is the sphere): a triangle, both sides hit, with its
// geometric normal normalize((v1 - v0) x (v2 - v0)).  Triangle k of a scene is primitive
// <number of spheres> + k for the lights' primId.
class Triangle {
public:
    Triangle(const Point &a, const Point &b, const Point &c, uint32_t matId, int lightId_ = -1)
        : v0(a), v1(b), v2(c), materialId(matId), lightId(lightId_) {}
    bool IsLight() const { return lightId > -1; }
    Point v0, v1, v2;
    uint32_t materialId;
    int lightId;
};

enum LightType { POINT_LIGHT = PT_POINT_LIGHT, AREA_LIGHT = PT_AREA_LIGHT };

// light.h:36-53
class Light {
public:
    Light(const Point &p, const Color &l) : type(POINT_LIGHT), pos(p), primId(0), intensity(l) {}
    Light(uint32_t pId, const Color &l) : type(AREA_LIGHT), pos(), primId(pId), intensity(l) {}
    LightType type;
    Point pos;
    uint32_t primId;
    Color intensity;
};

// pathtracer.h:38-65
class Pathtracer {
public:
    Pathtracer(Camera *cam, int nBounces, uint64_t seed = 1) : camera_(cam), handle_(nullptr)
    {
        if (pt_create(&handle_, &cam->Pod(), nBounces, seed) != PT_OK)
            throw std::runtime_error(std::string("Pathtracer: ") + pt_last_error(nullptr));
    }
    ~Pathtracer() { Release(); }
    Pathtracer(const Pathtracer &) = delete;
    Pathtracer &operator=(const Pathtracer &) = delete;

    // pathtracer.cu:172-220
    bool Init(const std::vector<Material> &mv, const std::vector<Primitive> &pv, const std::vector<Light> &lv)
    {
        return Init(mv, pv, std::vector<Triangle>(), lv);
    }
    // The same with triangles beside the spheres (not in the reference).
    bool Init(const std::vector<Material> &mv, const std::vector<Primitive> &pv, const std::vector<Triangle> &tv,
              const std::vector<Light> &lv)
    {
        std::vector<pt_material> m(mv.size());
        std::vector<pt_sphere> p(pv.size());
        std::vector<pt_light> l(lv.size());
        for (size_t i = 0; i < mv.size(); ++i) {
            m[i].color[0] = mv[i].color.r; m[i].color[1] = mv[i].color.g; m[i].color[2] = mv[i].color.b;
            m[i].coef = mv[i].coef; m[i].type = (int32_t)mv[i].type;
        }
        for (size_t i = 0; i < pv.size(); ++i) {
            p[i].pos[0] = pv[i].pos.x; p[i].pos[1] = pv[i].pos.y; p[i].pos[2] = pv[i].pos.z;
            p[i].radius = pv[i].radius; p[i].material_id = pv[i].materialId; p[i].light_id = pv[i].lightId;
        }
        for (size_t i = 0; i < lv.size(); ++i) {
            l[i] = pt_light{};
            l[i].type = (int32_t)lv[i].type;
            if (lv[i].type == AREA_LIGHT) l[i].u.prim_id = lv[i].primId;
            else { l[i].u.pos[0] = lv[i].pos.x; l[i].u.pos[1] = lv[i].pos.y; l[i].u.pos[2] = lv[i].pos.z; }
            l[i].intensity[0] = lv[i].intensity.r; l[i].intensity[1] = lv[i].intensity.g; l[i].intensity[2] = lv[i].intensity.b;
        }
        std::vector<pt_triangle> t(tv.size());
        for (size_t i = 0; i < tv.size(); ++i) {
            const Point *v[3] = {&tv[i].v0, &tv[i].v1, &tv[i].v2};
            float *dst[3] = {t[i].v0, t[i].v1, t[i].v2};
            for (int k = 0; k < 3; ++k) { dst[k][0] = v[k]->x; dst[k][1] = v[k]->y; dst[k][2] = v[k]->z; }
            t[i].material_id = tv[i].materialId; t[i].light_id = tv[i].lightId;
        }
        return pt_init_scene_triangles(handle_, m.data(), (uint32_t)m.size(), p.data(), (uint32_t)p.size(),
                                       t.data(), (uint32_t)t.size(), l.data(), (uint32_t)l.size()) == PT_OK;
    }

    // pathtracer.cu:222-243.  The reference re-reads *camera on every Run (pathtracer.cu:232).
    void Run(float *devBuffer)
    {
        check(pt_set_camera(handle_, &camera_->Pod()));
        check(pt_run(handle_, devBuffer, 1));
    }
    void Reset() { check(pt_reset(handle_)); }
    uint32_t GetIteration() const { return pt_iteration(handle_); }
    // Not in the reference: n Run() calls in one launch, checkpoint files, nearest-hit search mode.
    void RunN(float *devBuffer, uint32_t n, bool sync = true)
    {
        check(pt_set_camera(handle_, &camera_->Pod()));
        check(pt_run_n(handle_, devBuffer, n, sync ? 1 : 0));
    }
    // Run without the wait: consecutive calls overlap on the device, Wait() ends the series
    // (pt_run_async / pt_wait).  Run itself stays blocking, as the reference's is (pathtracer.cu:241).
    void RunAsync(float *devBuffer, uint32_t n = 1)
    {
        check(pt_set_camera(handle_, &camera_->Pod()));
        check(pt_run_async(handle_, devBuffer, n));
    }
    void Wait() { check(pt_wait(handle_)); }
    void SaveCheckpoint(const float *devBuffer, const std::string &path) { check(pt_save_checkpoint(handle_, devBuffer, path.c_str())); }
    void LoadCheckpoint(float *devBuffer, const std::string &path) { check(pt_load_checkpoint(handle_, devBuffer, path.c_str())); }
    void SetAccel(int mode) { check(pt_set_accel(handle_, mode)); }
    // Sharding plumbing (pt_set_stream, pt_set_accum_mode, pt_set_iteration, pt_set_row_bands, pt_ray_count).
    void SetStream(void *hipStream) { check(pt_set_stream(handle_, hipStream)); }
    void SetAccumMode(int mode) { check(pt_set_accum_mode(handle_, mode)); }
    void SetIteration(uint32_t iteration) { check(pt_set_iteration(handle_, iteration)); }
    void SetRowBands(uint32_t bandRows, uint32_t rank, uint32_t world) { check(pt_set_row_bands(handle_, bandRows, rank, world)); }
    uint64_t RayCount() const { uint64_t v = 0; check(pt_ray_count(handle_, &v)); return v; }
    void Release()
    {
        if (handle_) { pt_destroy(handle_); handle_ = nullptr; }
    }
    const char *LastError() const { return pt_last_error(handle_); }
    pt_handle Handle() const { return handle_; }

private:
    void check(int rc) const
    {
        if (rc != PT_OK) throw std::runtime_error(std::string("Pathtracer: ") + pt_last_error(handle_));
    }
    Camera *camera_;
    pt_handle handle_;
};
