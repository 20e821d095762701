// Host-side scene helpers of libptmi: camera parameters and built-in scenes.
// Plain host code (no device needed); fp32 in the reference's evaluation order, built with
// -ffp-contract=off like everything else.
#include "ptmi.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct H3 { float x, y, z; };
inline H3 h3(float x, float y, float z) { return H3{x, y, z}; }
inline H3 operator-(H3 a, H3 b) { return h3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline H3 operator+(H3 a, H3 b) { return h3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline H3 operator*(H3 a, float d) { return h3(a.x * d, a.y * d, a.z * d); }
inline float hdot(H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline H3 hcross(H3 a, H3 b) { return h3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float hlength(H3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
inline H3 hdiv(H3 a, float d) { const float inv = 1.f / d; return a * inv; } // geometry.h:124-126
inline H3 hnormalize(H3 a) { return hdiv(a, hlength(a)); }

// Philox4x32-10 for procedural scene generation (same generator the kernels use).
void philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
inline float u01(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }

pt_material make_material(int type, float r, float g, float b, float coef)
{
    pt_material m;
    m.color[0] = r; m.color[1] = g; m.color[2] = b;
    m.coef = coef;
    m.type = type;
    return m;
}

pt_sphere make_sphere(float x, float y, float z, float radius, uint32_t mat, int32_t light = -1)
{
    pt_sphere s;
    s.pos[0] = x; s.pos[1] = y; s.pos[2] = z;
    s.radius = radius;
    s.material_id = mat;
    s.light_id = light;
    return s;
}

struct SceneData {
    std::vector<pt_material> mats;
    std::vector<pt_sphere> prims;
    std::vector<pt_light> lights;
};

// Materials 0..5 of InitScene, main.cpp:133-138.
void demo_materials(SceneData &s)
{
    s.mats.push_back(make_material(PT_DIFFUSE, 1.f, 1.f, 1.f, 0.f));
    s.mats.push_back(make_material(PT_DIFFUSE, 0.75f, 0.25f, 0.25f, 0.f));
    s.mats.push_back(make_material(PT_DIFFUSE, 0.25f, 0.25f, 0.75f, 0.f));
    s.mats.push_back(make_material(PT_DIFFUSE, 0.75f, 0.75f, 0.75f, 0.f));
    s.mats.push_back(make_material(PT_SPECULAR, 0.999f, 0.999f, 0.999f, 1.f));
    s.mats.push_back(make_material(PT_TRANSMISSIVE, 0.999f, 0.999f, 0.999f, 1.5f));
}

// The six wall spheres of "scene 2", main.cpp:153-158.  The literals are double expressions
// narrowed to float at the Point/float parameters, as in the reference.
void demo_walls(SceneData &s)
{
    s.prims.push_back(make_sphere((float)(1e5 + 50), 40.f, 0.f, (float)1e5, 2));
    s.prims.push_back(make_sphere((float)(-1e5 - 50), 40.f, 0.f, (float)1e5, 1));
    s.prims.push_back(make_sphere(0.f, 40.f, (float)(-1e5 - 80), (float)1e5, 3));
    s.prims.push_back(make_sphere(0.f, 40.f, (float)(1e5 + 80), (float)1e5, 3));
    s.prims.push_back(make_sphere(0.f, (float)-1e5, 0.f, (float)1e5, 3));
    s.prims.push_back(make_sphere(0.f, (float)(1e5 + 80), 0.f, (float)1e5, 3));
}

// The area light, main.cpp:161,164; `index` is the sphere's position in the primitive list.
void demo_light(SceneData &s)
{
    const uint32_t index = (uint32_t)s.prims.size();
    s.prims.push_back(make_sphere(0.f, (float)579.6, -40.f, 500.f, 1, (int32_t)s.lights.size()));
    pt_light l;
    memset(&l, 0, sizeof l);
    l.type = PT_AREA_LIGHT;
    l.u.prim_id = index;
    l.intensity[0] = l.intensity[1] = l.intensity[2] = 12.f;
    s.lights.push_back(l);
}

void build_demo(SceneData &s)
{
    demo_materials(s);
    demo_walls(s);
    s.prims.push_back(make_sphere(-25.f, 16.5f, -50.f, 16.5f, 4)); // main.cpp:159
    s.prims.push_back(make_sphere(25.f, 16.5f, -25.f, 16.5f, 5));  // main.cpp:160
    demo_light(s);
}

// Demo room (walls + light first, so tie-breaks between them stay as in the demo scene) filled
// with 10 000 small spheres: 70 % diffuse with a random albedo, 15 % mirror, 15 % glass.
void build_synth10k(SceneData &s, uint64_t seed)
{
    demo_materials(s);
    demo_walls(s);
    demo_light(s);
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (uint32_t i = 0; i < 10000; ++i) {
        uint32_t a[4], b[4];
        philox(i, 0, 0, 0x5ce4e, k0, k1, a);
        philox(i, 1, 0, 0x5ce4e, k0, k1, b);
        const float x = -45.f + 90.f * u01(a[0]);
        const float y = 5.f + 70.f * u01(a[1]);
        const float z = -75.f + 95.f * u01(a[2]);
        const float radius = 0.3f + 0.9f * u01(a[3]);
        const float pick = u01(b[0]);
        uint32_t mat;
        if (pick < 0.70f) {
            mat = (uint32_t)s.mats.size();
            s.mats.push_back(make_material(PT_DIFFUSE, 0.2f + 0.7f * u01(b[1]), 0.2f + 0.7f * u01(b[2]),
                                           0.2f + 0.7f * u01(b[3]), 0.f));
        } else if (pick < 0.85f) {
            mat = 4;
        } else {
            mat = 5;
        }
        s.prims.push_back(make_sphere(x, y, z, radius, mat));
    }
}

// Demo room + an 8x8 grid of spheres cycling diffuse / mirror / glass: every wave meets all
// three BSDF branches, and depth-16 paths reach the Russian-roulette stage.
void build_mixed(SceneData &s)
{
    demo_materials(s);
    demo_walls(s);
    demo_light(s);
    const uint32_t cycle[3] = { 1, 4, 5 };
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 8; ++i) {
            const float x = -38.5f + 11.f * (float)i;
            const float z = -72.f + 11.f * (float)j;
            const float y = 4.5f + 9.f * (float)((i + j) & 3);
            uint32_t mat = cycle[(i + 2 * j) % 3];
            if (mat == 1) mat = 1 + (uint32_t)((i * 3 + j) % 3); // red / blue / grey diffuse
            s.prims.push_back(make_sphere(x, y, z, 4.5f, mat));
        }
}

// Camera::UpdateParameter, camera.h:144-158; Radians, globals.h:95-97
void update_parameter(pt_camera_state *c)
{
    const H3 u = h3(c->view[0][0], c->view[0][1], c->view[0][2]);
    const H3 v = h3(c->view[1][0], c->view[1][1], c->view[1][2]);
    const H3 w = h3(c->view[2][0], c->view[2][1], c->view[2][2]);
    const float m_pi = 3.14159265358979323846f;
    const uint32_t width = c->pod.width, height = c->pod.height;
    float magnitude = c->dist * 2.f * tanf((m_pi / 180.f) * (c->hfov * 0.5f)) / (float)width;
    const H3 pxX = u * magnitude;
    magnitude = c->dist * 2.f * tanf((m_pi / 180.f) * (c->vfov * 0.5f)) / (float)height;
    const H3 pxY = v * magnitude;
    const H3 offset = pxY * (float)((float)height * 0.5f) - pxX * (float)((float)width * 0.5f);
    const H3 first = w * c->dist + offset;
    c->pod.px_x[0] = pxX.x; c->pod.px_x[1] = pxX.y; c->pod.px_x[2] = pxX.z;
    c->pod.px_y[0] = pxY.x; c->pod.px_y[1] = pxY.y; c->pod.px_y[2] = pxY.z;
    c->pod.first_ray_dir[0] = first.x; c->pod.first_ray_dir[1] = first.y; c->pod.first_ray_dir[2] = first.z;
}

// Matrix::operator*, geometry.h:249-258, on 4x4 matrices whose last row/column are (0,0,0,1)
void mat_mul(const float a[4][4], const float b[4][4], float r[4][4])
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j] + a[i][3] * b[3][j];
}

} // namespace

extern "C" {

int pt_camera_state_init(pt_camera_state *cam, const float eye[3], const float look_at[3],
                         const float up_[3], uint32_t width, uint32_t height, float fov)
{
    if (!cam || !eye || !look_at || !up_ || width == 0 || height == 0) return PT_ERR_INVALID;
    // Camera::Camera, camera.h:31-57
    const float aspect = (float)height / (float)width;
    cam->hfov = fov;
    cam->vfov = cam->hfov * aspect;
    const H3 e = h3(eye[0], eye[1], eye[2]), up = h3(up_[0], up_[1], up_[2]);
    const H3 lookAt = h3(look_at[0], look_at[1], look_at[2]) - e;
    cam->dist = hlength(lookAt);
    const H3 w = hdiv(lookAt, cam->dist);
    const H3 v = hnormalize(up - w * hdot(up, w));
    const H3 u = hnormalize(hcross(w, v));
    cam->view[0][0] = u.x; cam->view[0][1] = u.y; cam->view[0][2] = u.z;
    cam->view[1][0] = v.x; cam->view[1][1] = v.y; cam->view[1][2] = v.z;
    cam->view[2][0] = w.x; cam->view[2][1] = w.y; cam->view[2][2] = w.z;
    cam->pod.pos[0] = e.x; cam->pod.pos[1] = e.y; cam->pod.pos[2] = e.z;
    cam->pod.width = width; cam->pod.height = height;
    update_parameter(cam);
    return PT_OK;
}

int pt_camera_translate(pt_camera_state *cam, const float delta[3])
{
    if (!cam || !delta) return PT_ERR_INVALID;
    // camera.h:79-90: pos += u*delta.x + v*delta.y + w*delta.z; the image plane does not change
    const H3 u = h3(cam->view[0][0], cam->view[0][1], cam->view[0][2]);
    const H3 v = h3(cam->view[1][0], cam->view[1][1], cam->view[1][2]);
    const H3 w = h3(cam->view[2][0], cam->view[2][1], cam->view[2][2]);
    const H3 step = (u * delta[0] + v * delta[1]) + w * delta[2];
    cam->pod.pos[0] += step.x; cam->pod.pos[1] += step.y; cam->pod.pos[2] += step.z;
    return PT_OK;
}

int pt_camera_rotate(pt_camera_state *cam, const float theta[3])
{
    if (!cam || !theta) return PT_ERR_INVALID;
    // camera.h:97-129 -- note the reference's naming: the "x-axis" matrix uses theta.y and the
    // "y-axis" matrix theta.x; theta.z is ignored
    const float ctx = theta[0] == 0.f ? 1.f : cosf(theta[0]);
    const float cty = theta[1] == 0.f ? 1.f : cosf(theta[1]);
    const float stx = theta[0] == 0.f ? 0.f : sinf(theta[0]);
    const float sty = theta[1] == 0.f ? 0.f : sinf(theta[1]);
    const float rx[4][4] = { {1.f, 0.f, 0.f, 0.f}, {0.f, cty, -sty, 0.f}, {0.f, sty, cty, 0.f}, {0.f, 0.f, 0.f, 1.f} };
    const float ry[4][4] = { {ctx, 0.f, stx, 0.f}, {0.f, 1.f, 0.f, 0.f}, {-stx, 0.f, ctx, 0.f}, {0.f, 0.f, 0.f, 1.f} };
    float view[4][4] = { {cam->view[0][0], cam->view[0][1], cam->view[0][2], 0.f},
                         {cam->view[1][0], cam->view[1][1], cam->view[1][2], 0.f},
                         {cam->view[2][0], cam->view[2][1], cam->view[2][2], 0.f},
                         {0.f, 0.f, 0.f, 1.f} };
    float out[4][4];
    if (theta[0] == 0.f) {
        mat_mul(rx, view, out);
    } else if (theta[1] == 0.f) {
        mat_mul(ry, view, out);
    } else {
        float rxy[4][4];
        mat_mul(rx, ry, rxy);
        mat_mul(rxy, view, out);
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) cam->view[i][j] = out[i][j];
    update_parameter(cam);
    return PT_OK;
}

int pt_camera_init(pt_camera_pod *out, const float eye[3], const float look_at[3],
                   const float up_[3], uint32_t width, uint32_t height, float fov)
{
    // Camera::Camera + UpdateParameter (camera.h:31-57,144-158): the POD of the full camera state
    if (!out) return PT_ERR_INVALID;
    pt_camera_state state;
    const int rc = pt_camera_state_init(&state, eye, look_at, up_, width, height, fov);
    if (rc == PT_OK) *out = state.pod;
    return rc;
}

int pt_scene_build(int scene_id, uint64_t seed,
                   pt_material *materials, uint32_t *n_materials,
                   pt_sphere *spheres, uint32_t *n_spheres,
                   pt_light *lights, uint32_t *n_lights,
                   float eye[3], float look_at[3], float up[3], float *fov_deg)
{
    if (!n_materials || !n_spheres || !n_lights) return PT_ERR_INVALID;
    SceneData s;
    switch (scene_id) {
    case PT_SCENE_DEMO: build_demo(s); break;
    case PT_SCENE_SYNTH10K: build_synth10k(s, seed); break;
    case PT_SCENE_MIXED: build_mixed(s); break;
    default: return PT_ERR_INVALID;
    }
    // camera of InitScene, main.cpp:129
    if (eye) { eye[0] = 0.f; eye[1] = 45.f; eye[2] = 79.5f; }
    if (look_at) { look_at[0] = 0.f; look_at[1] = 35.f; look_at[2] = 0.f; }
    if (up) { up[0] = 0.f; up[1] = 1.f; up[2] = 0.f; }
    if (fov_deg) *fov_deg = 60.f;

    const bool query = !materials && !spheres && !lights;
    if (!query) {
        if (!materials || !spheres || !lights) return PT_ERR_INVALID;
        if (*n_materials < s.mats.size() || *n_spheres < s.prims.size() || *n_lights < s.lights.size())
            return PT_ERR_INVALID;
        memcpy(materials, s.mats.data(), s.mats.size() * sizeof(pt_material));
        memcpy(spheres, s.prims.data(), s.prims.size() * sizeof(pt_sphere));
        memcpy(lights, s.lights.data(), s.lights.size() * sizeof(pt_light));
    }
    *n_materials = (uint32_t)s.mats.size();
    *n_spheres = (uint32_t)s.prims.size();
    *n_lights = (uint32_t)s.lights.size();
    return PT_OK;
}

} // extern "C"
