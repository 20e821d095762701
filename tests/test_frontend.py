"""The headless front end (examples/pt_headless, the replacement for the reference's main.cpp):
scene description files on the CPU, rendering / image output / checkpoints on the GPU."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

EXE = os.path.join(ROOT, "examples", "pt_headless")


@pytest.fixture(scope="module")
def exe():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", ROOT, "examples"], check=True, capture_output=True)
    return EXE


def run(exe, *args, check=True):
    p = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=300)
    if check:
        assert p.returncode == 0, p.stderr + p.stdout
    return p


def parse_dump(text):
    cam, mats, sph, lights = None, [], [], []
    for line in text.splitlines():
        f = line.split()
        if f[0] == "camera":
            cam = [np.float32(x) for x in f[1:]]
        elif f[0] == "material":
            mats.append((f[1], [np.float32(x) for x in f[2:]]))
        elif f[0] == "sphere":
            sph.append(([np.float32(x) for x in f[1:5]], int(f[5])))
        elif f[0] == "light":
            lights.append((int(f[1]), [np.float32(x) for x in f[2:]]))
    return cam, mats, sph, lights


@pytest.mark.parametrize("name,scene_id", [("demo", "SCENE_DEMO"), ("mixed", "SCENE_MIXED"), ("synth10k", "SCENE_SYNTH10K")])
def test_dump_scene_matches_the_library_scene(exe, pt, name, scene_id):
    """`--scene demo` is InitScene written out in the front end (main.cpp:128-166); it must be the
    same scene, float for float, as pt_scene_build's, which the golden vectors were made from."""
    cam, mats, sph, lights = parse_dump(run(exe, "--scene", name, "--dump-scene").stdout)
    m, s, l, (eye, look, up, fov) = pt.build_scene(getattr(pt, scene_id), 1)
    assert np.array_equal(np.array(cam, np.float32), np.array([*eye, *look, *up, fov], np.float32))
    assert len(mats) == len(m) and len(sph) == len(s) and len(lights) == len(l)
    kinds = {"diffuse": 0, "specular": 1, "transmissive": 2}
    for (kind, v), ref in zip(mats, m):
        assert kinds[kind] == ref["type"]
        assert np.array_equal(np.array(v, np.float32), np.array([*ref["color"], ref["coef"]], np.float32))
    for i, ((v, mat), ref) in enumerate(zip(sph, s)):
        assert mat == ref["material_id"]
        assert np.array_equal(np.array(v, np.float32), np.array([*ref["pos"], ref["radius"]], np.float32)), i
    for k, ((prim, v), ref) in enumerate(zip(lights, l)):
        assert prim == ref["u_bits"][0] and s[prim]["light_id"] == k
        assert np.array_equal(np.array(v, np.float32), ref["intensity"])


def test_scene_file_round_trip(exe, tmp_path):
    text = run(exe, "--scene", "mixed", "--dump-scene").stdout
    path = tmp_path / "mixed.scene"
    lines = text.splitlines()
    lines[3] += "   # trailing comment"
    path.write_text("# a comment\n\n" + "\n".join(lines) + "\n")
    assert run(exe, "--scene", path, "--dump-scene").stdout == text


def test_scene_file_with_triangles_round_trips(exe, pt, tmp_path):
    """Triangles (this build's extension) in the text format: statements in any order come back
    spheres first, and a triangle's light names primitive <spheres> + <triangle> to the library."""
    from triangle_scenes import box_room, scene_file_text
    mats, sph, tris, lights, camera = box_room(pt)
    text = scene_file_text(mats, sph, tris, lights, camera)
    path = tmp_path / "room.scene"
    path.write_text(text)
    once = run(exe, "--scene", path, "--dump-scene").stdout
    assert sum(l.startswith("triangle ") for l in once.splitlines()) == len(tris)
    assert "trianglelight 12 12 12 12" in once and "trianglelight 13 12 12 12" in once
    # triangle statements ahead of the spheres: the same scene
    lines = text.splitlines()
    shuffled = [lines[0]] + [l for l in lines if l.startswith("material")] + [l for l in lines if l.startswith("tri")] \
        + [l for l in lines if l.startswith("sphere")]
    path.write_text("\n".join(shuffled) + "\n")
    assert run(exe, "--scene", path, "--dump-scene").stdout == once
    path.write_text(once)
    assert run(exe, "--scene", path, "--dump-scene").stdout == once


@pytest.mark.gpu
def test_cli_triangle_scene_renders_like_the_python_mirror(exe, pt, tmp_path):
    import torch
    from triangle_scenes import box_room, scene_file_text
    mats, sph, tris, lights, camera = box_room(pt)
    scene = tmp_path / "room.scene"
    scene.write_text(scene_file_text(mats, sph, tris, lights, camera))
    out = tmp_path / "room.pfm"
    run(exe, "--scene", scene, "--width", 96, "--height", 64, "--depth", 6, "--seed", 5, "--iterations", 3, "--out", out)
    eye, look, up, fov = camera
    tr = pt.Pathtracer(pt.Camera(eye, look, up, 96, 64, 0, 0, fov), 6, 5)
    tr.Init(mats, sph, lights, tris)
    buf = torch.zeros(96 * 64 * 3, dtype=torch.float32, device="cuda")
    tr.RunN(buf, 3)
    torch.cuda.synchronize()
    assert read_pfm(out).tobytes() == buf.cpu().numpy().tobytes()
    assert float(buf.max()) > 0.0
    tr.Release()


@pytest.mark.parametrize("body,needle", [
    ("material diffuse 1 1 1\nsphere 0 0 0 1 1\n", "line 2: sphere refers to material 1"),
    ("material shiny 1 1 1\n", "line 1: unknown material type 'shiny'"),
    ("material diffuse 1 1 1\nsphere 0 0 0 -1 0\n", "line 2: sphere radius must be positive"),
    ("material diffuse 1 1 1\nsphere 0 0 0 1 0\nlight 3 1 1 1\n", "line 3: light refers to sphere 3"),
    ("material diffuse 1 1 1\nsphere 0 0 0 1 0\nlight 0 1 1 1\nlight 0 1 1 1\n", "line 4: sphere 0 already is a light"),
    ("material diffuse 1 1 1\nsphere 0 0 0 1 0 extra\n", "line 2: sphere needs"),
    ("material diffuse 1 nan 1\n", "line 1: material needs"),
    ("camera 0 0 0 0 0 1 0 1 0 190\n", "line 1: fov must be"),
    ("quad 0 0 0\n", "line 1: unknown statement 'quad'"),
    ("material diffuse 1 1 1\ntriangle 0 0 0 1 0 0 0 1\n", "line 2: triangle needs"),
    ("material diffuse 1 1 1\ntriangle 0 0 0 1 0 0 0 1 0 2\n", "line 2: triangle refers to material 2"),
    ("material diffuse 1 1 1\ntriangle 0 0 0 1 0 0 0 1 0 0\ntrianglelight 1 1 1 1\n", "line 3: trianglelight refers to triangle 1"),
    ("material diffuse 1 1 1\ntriangle 0 0 0 1 0 0 0 1 0 0\ntrianglelight 0 1 1 1\ntrianglelight 0 1 1 1\n", "line 4: triangle 0 already is a light"),
    ("material diffuse 1 1 1\n", "the scene has no sphere and no triangle"),
])
def test_scene_file_errors_name_the_line(exe, tmp_path, body, needle):
    path = tmp_path / "bad.scene"
    path.write_text(body)
    p = run(exe, "--scene", path, "--dump-scene", check=False)
    assert p.returncode == 1 and needle in p.stderr, p.stderr


def test_usage_errors(exe, tmp_path):
    for args in (["--width", "0"], ["--depth", "-3"], ["--out", "x.png"], ["--accel", "maybe"], ["--nonsense", "1"],
                 ["--width"]):
        p = run(exe, *args, check=False)
        assert p.returncode == 2 and "usage:" in p.stderr
    p = run(exe, "--scene", tmp_path / "missing.scene", check=False)
    assert p.returncode == 1 and "cannot open" in p.stderr
    camless = tmp_path / "nocam.scene"
    camless.write_text("material diffuse 1 1 1\nsphere 0 0 0 1 0\n")
    p = run(exe, "--scene", camless, "--out", tmp_path / "x.pfm", check=False)
    assert p.returncode == 1 and "no camera statement" in p.stderr


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0          # little endian
        data = np.frombuffer(f.read(), "<f4").reshape(h, w, 3)
    return data[::-1]                            # stored bottom-up


@pytest.mark.gpu
def test_cli_render_matches_golden_and_resumes(exe, tmp_path):
    g = load_golden("demo_64x48_d8_it1-4")
    common = ["--width", 64, "--height", 48, "--depth", int(g["max_bounces"]), "--seed", int(g["seed"])]
    full = tmp_path / "full.pfm"
    p = run(exe, *common, "--iterations", 4, "--out", full)
    assert "sps : 4)" in p.stdout
    assert read_pfm(full).tobytes() == g["image"].tobytes()
    # one Run() per iteration, as the reference's Display() does
    single = tmp_path / "single.pfm"
    run(exe, *common, "--iterations", 4, "--batch", 1, "--out", single)
    assert single.read_bytes() == full.read_bytes()
    # 2 iterations + checkpoint, then resume to 4
    ckp, half, resumed = tmp_path / "r.ckp", tmp_path / "half.pfm", tmp_path / "resumed.pfm"
    run(exe, *common, "--iterations", 2, "--checkpoint", ckp, "--out", half)
    p = run(exe, *common, "--iterations", 4, "--resume", ckp, "--out", resumed)
    assert "resumed" in p.stdout and "sps : 4)" in p.stdout
    assert resumed.read_bytes() == full.read_bytes()
    assert half.read_bytes() != full.read_bytes()
    # a checkpoint of another render is refused
    p = run(exe, "--width", 64, "--height", 48, "--depth", 3, "--resume", ckp, "--out", resumed, check=False)
    assert p.returncode == 1 and "depth" in p.stderr
    # 8-bit output: gamma 2.2, clamped
    ppm = tmp_path / "full.ppm"
    run(exe, *common, "--iterations", 4, "--out", ppm)
    raw = ppm.read_bytes()
    assert raw.startswith(b"P6\n64 48\n255\n")
    px = np.frombuffer(raw[len(b"P6\n64 48\n255\n"):], np.uint8).reshape(48, 64, 3)
    c = np.clip(g["image"], 0, 1)
    expect = (np.power(c, np.float32(1 / 2.2), dtype=np.float32) * np.float32(255) + np.float32(0.5)).astype(np.uint8)
    assert np.abs(px.astype(int) - expect.astype(int)).max() <= 1     # powf vs numpy in the last bit


@pytest.mark.gpu
def test_cli_scene_file_renders_like_the_builtin_scene(exe, tmp_path):
    scene = tmp_path / "mixed.scene"
    scene.write_text(run(exe, "--scene", "mixed", "--dump-scene").stdout)
    a, b = tmp_path / "a.pfm", tmp_path / "b.pfm"
    g = load_golden("mixed_64x64_d16_it1-2")
    common = ["--width", 64, "--height", 64, "--depth", int(g["max_bounces"]), "--seed", int(g["seed"]), "--iterations", 2]
    run(exe, *common, "--scene", "mixed", "--out", a)
    run(exe, *common, "--scene", scene, "--out", b)
    assert a.read_bytes() == b.read_bytes()
    assert read_pfm(a).tobytes() == g["image"].tobytes()


def parse_plan(text):
    out = []
    for line in text.splitlines():
        f = line.split()
        assert f[0] == "rank" and f[2] == "first_iteration" and f[4] == "iterations" and f[6] == "rows"
        out.append((int(f[1]), int(f[3]), int(f[5]), [tuple(map(int, r.split("-"))) for r in f[7:]]))
    return out


@pytest.mark.parametrize("world,height,iterations", [(8, 2160, 1024), (8, 1080, 64), (3, 100, 64), (2, 16, 1), (5, 48, 3), (7, 1080, 1000), (8, 33, 9)])
def test_multi_gpu_partition_matches_the_python_sharding(exe, world, height, iterations):
    """`pt_headless --plan` (host/multi_gpu.hpp) against sharding.owned_bands / sample_range: the
    C++ and the Python front end cut a render the same way, every row and every iteration is
    rendered exactly once."""
    import importlib
    sharding = importlib.import_module("cuda-pathtracer_amd.sharding")
    common = ["--plan", "--gpus", world, "--height", height, "--iterations", iterations]
    tile = parse_plan(run(exe, *common, "--split", "tile").stdout)
    assert len(tile) == world
    seen = np.zeros(height, int)
    for rank, first, n, rows in tile:
        assert (first, n) == (1, iterations)
        assert rows == sharding.owned_bands(height, 16, rank, world)
        for y0, y1 in rows:
            seen[y0:y1] += 1
    assert (seen == 1).all()
    sample = parse_plan(run(exe, *common, "--split", "sample").stdout)
    covered = np.zeros(iterations + 1, int)
    for rank, first, n, rows in sample:
        assert rows == [(0, height)]
        covered[first:first + n] += 1
        assert (first, n) == sharding.sample_range(rank, world, iterations)
    assert (covered[1:] == 1).all() and covered[0] == 0
    sizes = [n for _, _, n, _ in sample]
    assert max(sizes) - min(sizes) <= 1


def test_multi_gpu_usage_errors(exe, tmp_path):
    for args in (["--gpus", "0"], ["--split", "rows"], ["--gpus", "2", "--checkpoint", tmp_path / "c.ckp"]):
        p = run(exe, *args, check=False)
        assert p.returncode == 2 and "usage:" in p.stderr


@pytest.mark.gpu
def test_cli_multi_gpu_path_on_one_gpu(exe, tmp_path):
    """The sharded render path of the front end -- host thread per device, RCCL communicator,
    ncclReduce(sum) of the accumulation buffers -- with a communicator of one (all a one-GPU box
    can run): the tile split is the golden image bit for bit, the sample split sums in another
    order (rtol 2e-6 * sqrt(n) + atol 1e-6, as for PT_ACCUM_SUM elsewhere)."""
    g = load_golden("demo_64x48_d8_it1-4")
    common = ["--width", 64, "--height", 48, "--depth", int(g["max_bounces"]), "--seed", int(g["seed"]), "--iterations", 4,
              "--gpus", 1]
    tile, sample = tmp_path / "tile.pfm", tmp_path / "sample.pfm"
    p = run(exe, *common, "--split", "tile", "--out", tile)
    assert "on 1 GPU(s), tile split" in p.stdout
    assert read_pfm(tile).tobytes() == g["image"].tobytes()
    run(exe, *common, "--split", "sample", "--batch", 3, "--out", sample)
    np.testing.assert_allclose(read_pfm(sample), g["image"], rtol=2e-6 * 2, atol=1e-6)
    # a device that fails before the closing reduce: the first error comes back, nobody waits in the collective
    p = run(exe, *common, "--split", "sample", "--inject-failure", 0, "--out", sample, check=False)
    assert p.returncode == 1 and "device 0: injected failure before the reduce" in p.stderr
    # more devices than the box has: a clean error, no crash
    p = run(exe, *common[:-1], 64, "--split", "tile", "--out", tile, check=False)
    assert p.returncode == 1 and "visible" in p.stderr
