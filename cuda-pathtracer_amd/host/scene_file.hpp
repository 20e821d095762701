// Scene description files and image output for the headless front end.
//
// The reference hard-codes its scene in InitScene ("TODO: configuration file parsing",
// main.cpp:126) and shows its frame buffer through a GL pixel-buffer object; this header gives
// the same scene types a text format and writes the frame buffer as PFM (raw float) or PPM
// (8-bit, gamma 2.2: the reference's unused GAMMA_FLOAT, pathtracer.cu:29).
//
// Scene file, one statement per line, '#' starts a comment, indices count from 0 in order of
// appearance:
//     camera   ex ey ez   lx ly lz   ux uy uz   fov_degrees
//     material diffuse|specular|transmissive   r g b   [coef]
//     sphere   x y z   radius   material_index
//     light    sphere_index   r g b            # the sphere emits (area light, light.h:43-45)
//     pointlight x y z   r g b                 # kept in the light list; Trace never reads it
//     triangle x0 y0 z0  x1 y1 z1  x2 y2 z2  material_index     # not in the reference; hit from both sides
//     trianglelight triangle_index   r g b     # the triangle emits on the side of (v1-v0) x (v2-v0)
// Triangles are primitives <number of spheres> + k to the library, whatever the order of the statements.
#pragma once

#include "pathtracer.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

struct SceneDescription {
    bool has_camera = false;
    Point eye, look_at;
    Vec up;
    float fov = 60.f;
    std::vector<Material> materials;
    std::vector<Primitive> primitives;
    std::vector<Triangle> triangles;
    std::vector<Light> lights;   // area lights on triangles: primId = primitives.size() + triangle index
};

// Returns an empty string on success, otherwise "line N: what is wrong".
inline std::string ParseScene(std::istream &in, SceneDescription *out)
{
    *out = SceneDescription();
    std::string line;
    int lineno = 0;
    std::vector<size_t> triangle_lights;   // lights whose primId still is a triangle index
    auto err = [&lineno](const std::string &what) { return "line " + std::to_string(lineno) + ": " + what; };
    while (std::getline(in, line)) {
        ++lineno;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        std::istringstream ss(line);
        std::string kw;
        if (!(ss >> kw)) continue;
        auto floats = [&ss](float *v, int n) {
            for (int i = 0; i < n; ++i)
                if (!(ss >> v[i]) || !std::isfinite(v[i])) return false;
            return true;
        };
        auto at_end = [&ss]() { std::string rest; return !(ss >> rest); };
        if (kw == "camera") {
            float v[10];
            if (!floats(v, 10) || !at_end()) return err("camera needs eye(3) lookat(3) up(3) fov");
            if (!(v[9] > 0.f && v[9] < 180.f)) return err("fov must be in (0, 180) degrees");
            out->eye = Point(v[0], v[1], v[2]);
            out->look_at = Point(v[3], v[4], v[5]);
            out->up = Vec(v[6], v[7], v[8]);
            out->fov = v[9];
            out->has_camera = true;
        } else if (kw == "material") {
            std::string type;
            float v[4] = {0, 0, 0, 0};
            if (!(ss >> type) || !floats(v, 3)) return err("material needs a type and r g b");
            if (!(ss >> v[3])) v[3] = type == "transmissive" ? 1.5f : (type == "specular" ? 1.f : 0.f);
            else if (!at_end()) return err("trailing text after material");
            if (type == "diffuse") out->materials.push_back(CreateDiffuseMaterial(Color(v[0], v[1], v[2]), v[3]));
            else if (type == "specular") out->materials.push_back(CreateSpecularMaterial(Color(v[0], v[1], v[2]), v[3]));
            else if (type == "transmissive") out->materials.push_back(CreateTransmissiveMaterial(Color(v[0], v[1], v[2]), v[3]));
            else return err("unknown material type '" + type + "'");
        } else if (kw == "sphere") {
            float v[4];
            long mat = -1;
            if (!floats(v, 4) || !(ss >> mat) || !at_end()) return err("sphere needs x y z radius material_index");
            if (!(v[3] > 0.f)) return err("sphere radius must be positive");
            if (mat < 0 || (size_t)mat >= out->materials.size()) return err("sphere refers to material " + std::to_string(mat) + ", which is not defined above");
            out->primitives.push_back(Primitive(Point(v[0], v[1], v[2]), v[3], (uint32_t)mat));
        } else if (kw == "light") {
            long prim = -1;
            float v[3];
            if (!(ss >> prim) || !floats(v, 3) || !at_end()) return err("light needs sphere_index r g b");
            if (prim < 0 || (size_t)prim >= out->primitives.size()) return err("light refers to sphere " + std::to_string(prim) + ", which is not defined above");
            if (out->primitives[(size_t)prim].IsLight()) return err("sphere " + std::to_string(prim) + " already is a light");
            out->primitives[(size_t)prim].lightId = (int)out->lights.size();
            out->lights.push_back(Light((uint32_t)prim, Color(v[0], v[1], v[2])));
        } else if (kw == "triangle") {
            float v[9];
            long mat = -1;
            if (!floats(v, 9) || !(ss >> mat) || !at_end()) return err("triangle needs three corners (x y z each) and material_index");
            if (mat < 0 || (size_t)mat >= out->materials.size()) return err("triangle refers to material " + std::to_string(mat) + ", which is not defined above");
            out->triangles.push_back(Triangle(Point(v[0], v[1], v[2]), Point(v[3], v[4], v[5]), Point(v[6], v[7], v[8]), (uint32_t)mat));
        } else if (kw == "trianglelight") {
            long tri = -1;
            float v[3];
            if (!(ss >> tri) || !floats(v, 3) || !at_end()) return err("trianglelight needs triangle_index r g b");
            if (tri < 0 || (size_t)tri >= out->triangles.size()) return err("trianglelight refers to triangle " + std::to_string(tri) + ", which is not defined above");
            if (out->triangles[(size_t)tri].IsLight()) return err("triangle " + std::to_string(tri) + " already is a light");
            out->triangles[(size_t)tri].lightId = (int)out->lights.size();
            triangle_lights.push_back(out->lights.size());
            out->lights.push_back(Light((uint32_t)tri, Color(v[0], v[1], v[2])));
        } else if (kw == "pointlight") {
            float v[6];
            if (!floats(v, 6) || !at_end()) return err("pointlight needs x y z r g b");
            out->lights.push_back(Light(Point(v[0], v[1], v[2]), Color(v[3], v[4], v[5])));
        } else {
            return err("unknown statement '" + kw + "'");
        }
    }
    if (out->primitives.empty() && out->triangles.empty()) { ++lineno; return err("the scene has no sphere and no triangle"); }
    for (size_t l : triangle_lights) out->lights[l].primId += (uint32_t)out->primitives.size();
    return std::string();
}

inline std::string LoadSceneFile(const std::string &path, SceneDescription *out)
{
    std::ifstream f(path);
    if (!f) return "cannot open " + path;
    const std::string e = ParseScene(f, out);
    return e.empty() ? e : path + ": " + e;
}

// The format back out (round-trips through ParseScene; %.9g keeps every float bit).
inline void WriteScene(FILE *f, const SceneDescription &s)
{
    if (s.has_camera)
        fprintf(f, "camera %.9g %.9g %.9g  %.9g %.9g %.9g  %.9g %.9g %.9g  %.9g\n", s.eye.x, s.eye.y, s.eye.z,
                s.look_at.x, s.look_at.y, s.look_at.z, s.up.x, s.up.y, s.up.z, s.fov);
    static const char *names[] = {"diffuse", "specular", "transmissive"};
    for (const Material &m : s.materials)
        fprintf(f, "material %s %.9g %.9g %.9g %.9g\n", names[(int)m.type], m.color.r, m.color.g, m.color.b, m.coef);
    for (const Primitive &p : s.primitives)
        fprintf(f, "sphere %.9g %.9g %.9g %.9g %u\n", p.pos.x, p.pos.y, p.pos.z, p.radius, p.materialId);
    for (const Triangle &t : s.triangles)
        fprintf(f, "triangle %.9g %.9g %.9g  %.9g %.9g %.9g  %.9g %.9g %.9g  %u\n", t.v0.x, t.v0.y, t.v0.z, t.v1.x, t.v1.y, t.v1.z,
                t.v2.x, t.v2.y, t.v2.z, t.materialId);
    for (const Light &l : s.lights) {
        if (l.type == AREA_LIGHT && l.primId >= s.primitives.size())
            fprintf(f, "trianglelight %u %.9g %.9g %.9g\n", l.primId - (uint32_t)s.primitives.size(), l.intensity.r, l.intensity.g, l.intensity.b);
        else if (l.type == AREA_LIGHT) fprintf(f, "light %u %.9g %.9g %.9g\n", l.primId, l.intensity.r, l.intensity.g, l.intensity.b);
        else fprintf(f, "pointlight %.9g %.9g %.9g %.9g %.9g %.9g\n", l.pos.x, l.pos.y, l.pos.z, l.intensity.r, l.intensity.g, l.intensity.b);
    }
}

// One of the library's built-in scenes (pt_scene_build) as a SceneDescription.
inline bool BuiltinScene(int which, uint64_t seed, SceneDescription *out)
{
    *out = SceneDescription();
    uint32_t nm = 0, ns = 0, nl = 0;
    float eye[3], look[3], up[3], fov;
    if (pt_scene_build(which, seed, nullptr, &nm, nullptr, &ns, nullptr, &nl, eye, look, up, &fov) != PT_OK) return false;
    std::vector<pt_material> m(nm);
    std::vector<pt_sphere> s(ns);
    std::vector<pt_light> l(nl);
    if (pt_scene_build(which, seed, m.data(), &nm, s.data(), &ns, l.data(), &nl, nullptr, nullptr, nullptr, nullptr) != PT_OK) return false;
    out->has_camera = true;
    out->eye = Point(eye[0], eye[1], eye[2]);
    out->look_at = Point(look[0], look[1], look[2]);
    out->up = Vec(up[0], up[1], up[2]);
    out->fov = fov;
    for (const pt_material &x : m) out->materials.push_back(MakeMaterial(Color(x.color[0], x.color[1], x.color[2]), x.coef, (MaterialType)x.type));
    for (const pt_sphere &x : s) out->primitives.push_back(Primitive(Point(x.pos[0], x.pos[1], x.pos[2]), x.radius, x.material_id, x.light_id));
    for (const pt_light &x : l) {
        const Color c(x.intensity[0], x.intensity[1], x.intensity[2]);
        if (x.type == PT_AREA_LIGHT) out->lights.push_back(Light(x.u.prim_id, c));
        else out->lights.push_back(Light(Point(x.u.pos[0], x.u.pos[1], x.u.pos[2]), c));
    }
    return true;
}

// rgb: height rows of width pixels, row 0 on top (the frame buffer's layout).
// PFM stores rows bottom-up, little-endian floats (negative scale).
inline bool WritePFM(const std::string &path, const float *rgb, uint32_t width, uint32_t height)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "PF\n%u %u\n-1.0\n", width, height);
    bool ok = true;
    for (uint32_t y = height; y-- > 0 && ok;)
        ok = fwrite(rgb + (size_t)y * width * 3, sizeof(float), (size_t)width * 3, f) == (size_t)width * 3;
    return fclose(f) == 0 && ok;
}

inline bool WritePPM(const std::string &path, const float *rgb, uint32_t width, uint32_t height, float gamma = 2.2f)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "P6\n%u %u\n255\n", width, height);
    std::vector<unsigned char> row((size_t)width * 3);
    bool ok = true;
    for (uint32_t y = 0; y < height && ok; ++y) {
        for (size_t i = 0; i < row.size(); ++i) {
            const float v = rgb[(size_t)y * width * 3 + i];
            const float c = v > 0.f ? (v < 1.f ? v : 1.f) : 0.f;   // NaN -> 0
            row[i] = (unsigned char)(powf(c, 1.f / gamma) * 255.f + 0.5f);
        }
        ok = fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    return fclose(f) == 0 && ok;
}
