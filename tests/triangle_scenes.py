"""Scenes with triangles for the tests (this build's extension: the reference has spheres only,
primitive.h:25-28).  Plain numpy records in the layouts of pt_material / pt_sphere / pt_triangle /
pt_light."""
import numpy as np


def box_room(pt):
    """The package's room of triangles: (materials, spheres, triangles, lights, camera tuple)."""
    return pt.triangle_room()


def triangle_soup(pt, rng, n_tris, n_spheres):
    """Random triangles and spheres of all three materials around the demo camera's view, with the
    cases a scan has to get right: degenerate triangles (no area), exact duplicates (equal distances:
    the lower index wins), slivers, a few very large and very small ones, and emissive triangles."""
    mats = np.zeros(5, pt.MATERIAL_DTYPE)
    mats["color"] = rng.uniform(0.2, 0.95, (5, 3))
    mats["type"] = [0, 1, 2, 0, 0]
    mats["coef"] = [0.0, 0.9, 1.4, 0.0, 0.0]
    tris = np.zeros(n_tris, pt.TRIANGLE_DTYPE)
    centre = rng.uniform(-30, 30, (n_tris, 3)) + np.array([0, 35, -20])
    size = rng.choice([0.5, 3.0, 12.0, 60.0], n_tris, p=[0.2, 0.5, 0.25, 0.05])[:, None]
    tris["v0"] = centre + rng.normal(size=(n_tris, 3)) * size
    tris["v1"] = centre + rng.normal(size=(n_tris, 3)) * size
    tris["v2"] = centre + rng.normal(size=(n_tris, 3)) * size
    tris["material_id"] = rng.integers(0, 5, n_tris)
    tris["light_id"] = -1
    if n_tris >= 8:
        tris["v2"][1] = tris["v1"][1]                 # degenerate: two equal vertices
        tris["v1"][2] = tris["v0"][2]; tris["v2"][2] = tris["v0"][2]    # a point
        tris[4] = tris[3]                             # exact duplicate
        tris["v2"][5] = tris["v0"][5] + (tris["v1"][5] - tris["v0"][5]) * np.float32(0.5)   # collinear
        tris["light_id"][6] = 0
        tris["light_id"][7] = 0
    sph = np.zeros(n_spheres, pt.SPHERE_DTYPE)
    sph["pos"] = rng.uniform(-30, 30, (n_spheres, 3)) + np.array([0, 35, -20])
    sph["radius"] = rng.uniform(1.0, 7.0, n_spheres) * (40.0 / max(n_spheres, 1)) ** (1 / 3)
    sph["material_id"] = rng.integers(0, 5, n_spheres)
    sph["light_id"] = -1
    if n_spheres:
        sph["light_id"][0] = 0
    lights = np.zeros(1, pt.LIGHT_DTYPE)
    lights["type"] = 1
    lights["u_bits"][0, 0] = 0 if n_spheres else min(6, n_tris - 1)
    lights["intensity"] = [[9, 8, 7]]
    return mats, sph, tris, lights, ([0, 45, 79.5], [0, 35, 0], [0, 1, 0], 55.0)


def scene_file_text(mats, sph, tris, lights, camera):
    """The scene in the headless front end's text format (host/scene_file.hpp); %.9g keeps every float bit."""
    g = lambda v: " ".join("%.9g" % float(x) for x in np.atleast_1d(v))
    eye, look, up, fov = camera
    out = ["camera %s  %s  %s  %s" % (g(np.float32(eye)), g(np.float32(look)), g(np.float32(up)), g(np.float32(fov)))]
    names = ["diffuse", "specular", "transmissive"]
    out += ["material %s %s %s" % (names[int(m["type"])], g(m["color"]), g(m["coef"])) for m in mats]
    out += ["sphere %s %s %d" % (g(s["pos"]), g(s["radius"]), int(s["material_id"])) for s in sph]
    out += ["triangle %s  %s  %s  %d" % (g(t["v0"]), g(t["v1"]), g(t["v2"]), int(t["material_id"])) for t in tris]
    # the text format gives every emitting primitive a light of its own; primitives that share one light
    # here (a quad light's two triangles) come out as lights of equal intensity, which the trace loop
    # cannot tell apart (it reads a light's intensity through the primitive that was hit, pathtracer.cu:133-137)
    for k, p in enumerate(sph):
        if p["light_id"] >= 0:
            out.append("light %d %s" % (k, g(lights[int(p["light_id"])]["intensity"])))
    for k, t in enumerate(tris):
        if t["light_id"] >= 0:
            out.append("trianglelight %d %s" % (k, g(lights[int(t["light_id"])]["intensity"])))
    return "\n".join(out) + "\n"
