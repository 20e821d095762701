#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload demo1080|demo4k|synth10k|mixed16|triroom1080]

A "step" is one full render of the workload: every pixel's camera rays, nearest-hit scans,
BSDF sampling and accumulation for all of the workload's iterations (one pt_run_n launch).
The metric is path-bounces per second: the number of Intersect calls (primary rays included,
scene.h:71-94), counted exactly by the kernel, divided by wall time with the frame buffer
resident in HBM.

N > 1: one process per GPU over RCCL.  Under torch.distributed.run (WORLD_SIZE set) this process
is one rank; typed as plain `python bench.py --gpus N` it starts the N ranks itself -- as a child
`python -m torch.distributed.run ... bench.py <same arguments>`, before this process has made any
GPU call -- and relays rank 0's JSON line.
  --split sample (default): every rank renders the same frame for its own block of iterations
      into a SUM buffer and one RCCL reduce(sum) of the accumulation buffers ends the step;
  --split tile: every rank renders its 16-row bands (band b -> rank b % N) for ALL world x
      iterations iterations, other pixels stay zero, the same reduce(sum) assembles the image,
      bit-identical to a one-GPU render.
Either way per-GPU work is fixed as N grows (weak scaling: the job is N x iterations iterations
of the frame).  Steps alternate between two buffers, so the reduce of step k (on RCCL's stream)
runs while step k+1 renders; the timed region ends when every reduce has completed.  The N > 1
line also carries `ranks_seen` (an all-reduce of ones) with the backend that carried it and `image_rmse_vs_single_rank`
(rank 0 re-renders a few row bands of the whole job alone and compares them with the reduced
frame).  --workload demo4k is the 8-GPU configuration's frame (3840x2160, 128 iterations per GPU).

Adds to the JSON line:
  other_workloads  (N = 1, default workload) short runs of the other configurations with the
                default kernels: 71 spheres at depth 16, 10 007 spheres through the BVH and by the
                linear scan, the 4K frame
  roofline      FP32 vector-ALU roofline of the render kernel: algorithmic flops per launch
                ((17*N_spheres + 121) per Intersect call) / HIP-event duration of the launch
  cpu_baseline  the CPU oracle (a C port of the reference's trace loop, pinned bit-exactly
                to the reference's host build) on the host's cores, bounded sample, rank 0, N=1
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# benchmark workloads: the 1080p demo render is the headline; the others are its 4K, 10k-sphere and mixed-material relatives
WORKLOADS = {
    "demo1080": dict(scene="SCENE_DEMO", width=1920, height=1080, iterations=64, depth=8,
                     label="demo scene (9 spheres + area light) 1920x1080, 64 iterations of 4 paths per pixel, depth 8"),
    # configs[2]: 3840x2160, 1024 spp over 8 GPUs under the sample-range split = 128 iterations per GPU
    "demo4k": dict(scene="SCENE_DEMO", width=3840, height=2160, iterations=128, depth=8,
                   label="demo scene (9 spheres + area light) 3840x2160, 128 iterations of 4 paths per pixel per GPU, depth 8"),
    "synth10k": dict(scene="SCENE_SYNTH10K", width=1920, height=1080, iterations=16, depth=8,
                     label="synthetic 10 007-sphere scene 1920x1080, 16 iterations of 4 paths per pixel, depth 8"),
    "mixed16": dict(scene="SCENE_MIXED", width=1920, height=1080, iterations=32, depth=16,
                    label="mixed diffuse/specular/glass grid 1920x1080, 32 iterations of 4 paths per pixel, depth 16"),
    # this build's triangle extension (no counterpart in the reference, parity unpinned): the demo room with walls of triangles
    "triroom1080": dict(scene="TRIANGLE_ROOM", width=1920, height=1080, iterations=64, depth=8,
                        label="demo room built from 14 triangles + 3 spheres (extension: the reference has no triangles) "
                              "1920x1080, 64 iterations of 4 paths per pixel, depth 8"),
}
PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X vector FP32 peak (256 CU x 256 flop/clk x 2.4 GHz)


def flops_per_ray(n_spheres: int, n_triangles: int = 0) -> int:
    """Algorithmic flops of one Intersect call + one bounce of shading: 17 per sphere test
    (3 sub, two 5-flop dots, b*b, -, r*r, +) + 21 for hit point/normal + 100 for the BSDF.
    A triangle test (Moeller-Trumbore, evaluated in full): two cross products of 9, four dot
    products of 5, o - v0, 1/det and three products with it = 45."""
    return 17 * n_spheres + 45 * n_triangles + 121


def load_scene(pt, wl):
    """(materials, spheres, triangles or None, lights, (eye, look_at, up, fov)) of a workload."""
    if wl["scene"] == "TRIANGLE_ROOM":
        mats, sph, tris, lights, cam = pt.triangle_room()
        return mats, sph, tris, lights, cam
    mats, sph, lights, cam = pt.build_scene(getattr(pt, wl["scene"]), 1)
    return mats, sph, None, lights, cam


def flops_per_ray_bvh(workload: str):
    """With the sphere BVH the work per Intersect call depends on the ray: the average visit
    counts come from the committed run of the counting build (tools/stats, pt_accel_stats):
    22 flops per child box of an inner node (6 fma + 10 min/max; four boxes per node), 4 sphere
    tests per leaf, plus the always-tested spheres and the bounce's shading.  None when no count
    is committed."""
    path = os.path.join(ROOT, "profiles", f"bvh_visits_{workload}.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        v = json.load(f)
    return (121 + 17 * v["always_tested_spheres"] + 22 * v.get("node_width", 2) * v["inner_per_ray"]
            + 4 * 17 * v["leaves_per_ray"])


def host_description():
    """The host the CPU baseline runs on: CPU model, the cores this process may use and the cores
    it does use.  On a shared GPU box the affinity mask shows the whole machine while the box's
    share per GPU is what OMP_NUM_THREADS says (16); the baseline uses every core of that share."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    affinity = len(os.sched_getaffinity(0))
    share = os.environ.get("OMP_NUM_THREADS", "")
    share = int(share) if share.isdigit() and int(share) > 0 else None
    used = max(1, min(affinity, share) if share else affinity)
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "cores_available_affinity": affinity,
            "cores_share_of_this_gpu": share, "cores_used": used}


def cpu_baseline(pt, wl, mats, sph, lights, cam, gpu_render=None, tris=None):
    """Time the CPU oracle on a bounded sample of the same workload (about 10-30 s of CPU work).
    gpu_render(n_iterations) -> numpy image: when given, the oracle's rows are also compared with
    the GPU's image of the same iterations (image RMSE / max abs difference; expected 0)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    host = host_description()
    threads = host["cores_used"]
    w, h = wl["width"], wl["height"]
    # ~15 s of CPU work: the oracle does about 5.4e6 Intersect calls/s/thread on 9 spheres
    # and scales with the number of sphere tests per call
    rate = 5.4e6 * 24.0 / (len(sph) + (2.5 * len(tris) if tris is not None else 0) + 15.0)
    budget_rays = 15.0 * threads * rate
    rays_per_row = w * 4 * 3.7
    if budget_rays >= rays_per_row * h:
        iters, rows = max(1, min(wl["iterations"], int(budget_rays / (rays_per_row * h)))), h
    else:
        iters, rows = 1, min(h, max(threads, int(budget_rays / rays_per_row)))
    y0 = (h - rows) // 2
    t0 = time.perf_counter()
    cpu_img, _, _, rays = O.oracle_render(cam.pod, mats, sph, lights, 1, wl["depth"], first_iteration=1,
                                          n_iterations=iters, rect=(0, y0, w, y0 + rows), n_threads=threads, triangles=tris)
    dt = time.perf_counter() - t0
    out = {"value": rays / dt, "unit": "path-bounces/s", "cores": threads, "kind": "port",
           "sample": f"{wl['label']}: rows {y0}..{y0 + rows} x {iters} iteration(s), {rays} Intersect calls "
                     f"in {dt:.2f} s on {threads} threads (oracle/pt_oracle.c)",
           "host": host}
    if gpu_render is not None:
        import numpy as np
        diff = gpu_render(iters)[y0:y0 + rows].astype(np.float64) - cpu_img[y0:y0 + rows]
        out["image_vs_gpu"] = {"rmse": float(np.sqrt((diff ** 2).mean())), "max_abs": float(np.abs(diff).max()),
                               "pixels": int(rows * w), "iterations": int(iters),
                               "note": "GPU image vs this CPU image, same seed and iterations; parity is bit-exact"}
    if O.have_ref() and tris is None:
        # the reference's own trace logic compiled for host (oracle/_ref, prebuilt in the
        # development container), single-threaded as the reference is, on a smaller sample
        rows1 = max(1, min(h, int(5.0 * rate / rays_per_row)))
        y1 = (h - rows1) // 2
        t0 = time.perf_counter()
        _, _, _, rays1 = O.ref_render(cam.eye, cam.look_at, cam.up, w, h, cam.fov, mats, sph, lights, 1,
                                      wl["depth"], rect=(0, y1, w, y1 + rows1))
        dt1 = time.perf_counter() - t0
        out["reference_host_build"] = {"value": rays1 / dt1, "unit": "path-bounces/s", "cores": 1,
                                       "kind": "reference",
                                       "sample": f"rows {y1}..{y1 + rows1} x 1 iteration, {rays1} Intersect "
                                                 f"calls in {dt1:.2f} s, 1 thread (oracle/_ref/libpt_ref.so)"}
    return out


LAUNCH_NAMES = None   # filled in main(): LAUNCH_* code -> (PMC file tag, kernel name)


def measure_other_workload(pt, torch, name, accel, steps):
    """One more workload on this GPU, outside the headline's timed region: K steps between two
    events on the render stream after one untimed step.  Returns its line for `other_workloads`."""
    wl = WORKLOADS[name]
    mats, sph, tris, lights, (eye, look, up, fov) = load_scene(pt, wl)
    cam = pt.Camera(eye, look, up, wl["width"], wl["height"], 0.0, 0.0, fov)
    tr = pt.Pathtracer(cam, wl["depth"], seed=1)
    try:
        tr.Init(mats, sph, lights, tris)
        tr.SetAccel(pt.ACCEL_AUTO if accel == "auto" else pt.ACCEL_BRUTE_FORCE)
        bvh = accel == "auto" and tr.AccelNodes() > 0
        tr.SetStream(torch.cuda.current_stream().cuda_stream)
        buf = torch.zeros(wl["width"] * wl["height"] * 3, dtype=torch.float32, device="cuda")
        tr.RunN(buf, wl["iterations"], sync=True)       # untimed: allocations, first launch
        tr.ResetCounters()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(steps):
            tr.Reset()
            tr.RunN(buf, wl["iterations"], sync=False)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / steps
        rays = tr.RayCount() / steps
        fpr = flops_per_ray_bvh(name) if bvh else flops_per_ray(len(sph), 0 if tris is None else len(tris))
        achieved = rays * fpr / (ms * 1e-3) / 1e12 if fpr else None
        return {"workload": wl["label"], "nearest_hit": "sphere BVH" if bvh else "linear scan", "steps": steps,
                "ms_per_step": ms, "value": rays / (ms * 1e-3), "unit": "path-bounces/s",
                "launch": LAUNCH_NAMES.get(tr.LastLaunch(), ("none", "none"))[1],
                "flops_per_bounce": fpr, "roofline_frac": achieved / PEAK_FP32_VALU_TFLOPS if achieved else None}
    finally:
        tr.Release()


def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as a child
    `python -m torch.distributed.run` of this same script.  This process has not touched the GPU
    (torch is not even imported here); it waits for the child, relays rank 0's JSON line on
    stdout and returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{") and '"metric"' in l]
    for l in proc.stdout.splitlines():
        if l not in lines:
            print(l, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    elif proc.returncode == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return proc.returncode


PMC_FIELDS = ("hbm_bytes_per_step", "valu_issue_util", "active_lane_frac", "wait_frac")


def pmc_summary(tag_path: str, build_id: str):
    """Issue-slot utilisation, active-lane fraction and HBM bytes of the render kernel from the
    committed PMC passes of this same command (profiles/pmc_*.json, written by
    tools/summarize_prof.py; rocprofv3 cannot be run from inside the benchmark).  The file carries
    the build id of the library the counters were taken from: for another build the counters are
    not quoted (all None) and `counters_from` says why.  Returns (counters, counters_from)."""
    rel = os.path.relpath(tag_path, ROOT)
    if not os.path.exists(tag_path):
        return {}, {"file": None, "note": f"no committed PMC pass for this launch ({rel})"}
    with open(tag_path) as f:
        v = json.load(f)
    source = {"file": rel, "tag": v.get("tag"), "build_id": v.get("build_id"), "this_build_id": build_id}
    if v.get("build_id") != build_id:
        source["stale"] = True
        source["note"] = "counters were taken from another build of libptmi.so: not quoted"
        return {}, source
    source["stale"] = False
    return {k: v[k] for k in PMC_FIELDS if k in v}, source


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="demo1080", choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=-1,
                    help="kernel variant (-1 auto: the path pool for scenes rendered by the linear scan, the wavefront "
                         "launches for BVH scenes; 0 regeneration megakernel, 1 classic megakernel, "
                         "100 wavefront, 500 work-stealing megakernel)")
    ap.add_argument("--accel", default="auto", choices=["auto", "brute"],
                    help="scenes above 192 spheres: sphere BVH (auto) or the reference's linear scan (brute)")
    ap.add_argument("--integrator", default="path", choices=["path", "nee"],
                    help="path: the reference's Trace (the headline metric); nee: opt-in direct-light sampling "
                         "with MIS (bounces then include the shadow rays)")
    ap.add_argument("--serial-launches", action="store_true",
                    help="profiling: one set of batch buffers on one stream (PT_TUNE_BUFFER_SETS 1), so that the "
                         "launches of a step run one after the other and a kernel trace of the step is additive")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="N = 1, default workload: do not append the short runs of the other workloads")
    ap.add_argument("--split", default="sample", choices=["sample", "tile"],
                    help="N > 1: sample-range split (each rank its own iterations, SUM buffers) or tile split "
                         "(each rank its own 16-row bands of all iterations); one RCCL reduce(sum) either way")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    # Libraries print banners on stdout (RCCL its version block, gloo its connection report):
    # everything but the result line goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # PT_BENCH_BACKEND=gloo is a functional rehearsal of the N>1 path on a box with fewer GPUs
    # than ranks (ranks share devices, the reduce is staged through host memory); the driver's
    # scaling runs use the default: one GPU per rank, RCCL.
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # PT_BENCH_SINGLE_RANK_PG=1: initialise the process group and issue the collectives even with
    # one rank (a rehearsal of the RCCL calls of the N>1 path on a one-GPU box)
    grouped = world > 1 or os.environ.get("PT_BENCH_SINGLE_RANK_PG") == "1"
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pt = importlib.import_module("cuda-pathtracer_amd")
    sharding = importlib.import_module("cuda-pathtracer_amd.sharding")
    wl = WORKLOADS[args.workload]
    mats, sph, tris, lights, (eye, look, up, fov) = load_scene(pt, wl)
    cam = pt.Camera(eye, look, up, wl["width"], wl["height"], 0.0, 0.0, fov)
    tr = pt.Pathtracer(cam, wl["depth"], seed=1)
    tr.Init(mats, sph, lights, tris)
    tr.SetKernelVariant(args.variant)
    tr.SetAccel(pt.ACCEL_AUTO if args.accel == "auto" else pt.ACCEL_BRUTE_FORCE)
    tr.SetIntegrator(pt.INTEGRATOR_NEE_MIS if args.integrator == "nee" else pt.INTEGRATOR_PATH)
    if args.serial_launches:
        tr.SetTuning(pt.TUNE_BUFFER_SETS, 1)
    bvh = args.accel == "auto" and tr.AccelNodes() > 0
    # what PT_VARIANT_AUTO resolves to is asked of the library after the warm-up (pt_last_launch)
    tr.SetStream(torch.cuda.current_stream().cuda_stream)
    tile_split = grouped and args.split == "tile"
    if grouped and not tile_split:
        tr.SetAccumMode(pt.ACCUM_SUM)
    if tile_split:
        tr.SetRowBands(16, rank, world)
    # N > 1: two accumulation buffers, so that the reduce of one step (on RCCL's stream) overlaps
    # with the render of the next step into the other buffer
    n_px = wl["width"] * wl["height"] * 3
    bufs = [torch.zeros(n_px, dtype=torch.float32, device=device) for _ in range(2 if grouped else 1)]
    buf = bufs[0]
    pending = [None] * len(bufs)
    if tile_split:   # every rank renders all world x iterations iterations of its own bands
        first, n_it = 1, wl["iterations"] * world
    else:
        first, n_it = sharding.sample_range(rank, world, wl["iterations"] * world)
    total_iterations = wl["iterations"] * world
    step_no = [0]

    def finish(slot):
        """The reduce that last used bufs[slot] is complete (as seen by the current stream) and the
        root has turned the sum into the mean."""
        if pending[slot] is not None:
            pending[slot].wait()
            if rank == 0 and not tile_split:
                bufs[slot].mul_(1.0 / float(total_iterations))
            pending[slot] = None

    def step():
        slot = step_no[0] % len(bufs)
        step_no[0] += 1
        finish(slot)
        if tile_split:
            bufs[slot].zero_()      # foreign bands must be zero (on the root they hold the last sum)
        tr.Reset()
        tr.SetIteration(first - 1)
        tr.RunN(bufs[slot], n_it, sync=False)
        if grouped:
            if backend == "nccl":
                pending[slot] = sharding.reduce_to_root(bufs[slot], dist, 0, async_op=True, single_rank_too=True)
            else:
                host = bufs[slot].cpu()
                sharding.reduce_to_root(host, dist, 0, single_rank_too=True)
                bufs[slot].copy_(host)
                if rank == 0 and not tile_split:
                    bufs[slot].mul_(1.0 / float(total_iterations))

    def fence():
        for slot in range(len(bufs)):
            finish(slot)
        torch.cuda.synchronize()
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def all_reduce_scalar(value, op):
        t = torch.tensor([value], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        if grouped:
            dist.all_reduce(t, op=op)
        return float(t.item())

    if grouped and backend == "nccl":
        # RCCL sets up its channels on a communicator's first collective of a kind: one untimed
        # reduce of a (still zero) frame-sized buffer, whatever --warmup says
        sharding.reduce_to_root(bufs[-1], dist, 0, single_rank_too=True)
    for _ in range(args.warmup):
        step()
    fence()
    tr.ResetCounters()
    # HIP events on the stream the kernels run on (the library was handed torch's current stream)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    dt = time.perf_counter() - t0
    region_ms = ev0.elapsed_time(ev1)
    rays_local = tr.RayCount()

    dt_max = all_reduce_scalar(dt, dist.ReduceOp.MAX)
    rays_total = all_reduce_scalar(float(rays_local), dist.ReduceOp.SUM)

    # N > 1 self-check: how many ranks the collective really spans, and the reduced frame of the
    # last step against rank 0 rendering a few row bands of the WHOLE job (all world x iterations
    # iterations) alone.  Outside the timed region.
    multi = None
    if grouped:
        grouped_launch_ms = tr.LastKernelMs()     # the last timed launch (before the check render below)
        ranks_seen = int(round(all_reduce_scalar(1.0, dist.ReduceOp.SUM)))
        names = [None] * world
        dist.all_gather_object(names, f"rank {rank}: cuda:{dev_index} {torch.cuda.get_device_name(dev_index)}")
        # ranks_seen counts the ranks of whatever backend carried the collective: RCCL only when backend is "nccl"
        multi = {"ranks_seen": ranks_seen, "backend": backend, "collective_library": "RCCL" if backend == "nccl" else backend,
                 "split": args.split, "ranks": names}
        if rank == 0:
            last = bufs[(step_no[0] - 1) % len(bufs)].clone()
            solo = torch.zeros_like(last)
            check_world = 68      # bands 0, 68, ...: rows 0-15 and 1088-1103 ... of the frame
            while check_world > 1 and (wl["height"] + 15) // 16 <= check_world:
                check_world //= 2
            tr.SetAccumMode(pt.ACCUM_RUNNING_MEAN)
            tr.SetRowBands(16, 0, check_world)
            tr.Reset()
            tr.RunN(solo, total_iterations, sync=True)
            rows = [r for (y0, y1) in sharding.owned_bands(wl["height"], 16, 0, check_world) for r in range(y0, y1)]
            a = last.view(wl["height"], wl["width"], 3)[rows].double()
            b = solo.view(wl["height"], wl["width"], 3)[rows].double()
            multi["image_rmse_vs_single_rank"] = float(((a - b) ** 2).mean().sqrt())
            multi["image_max_abs_vs_single_rank"] = float((a - b).abs().max())
            multi["image_check"] = (f"{len(rows)} rows x {total_iterations} iterations re-rendered by rank 0 alone "
                                    f"(running mean); mean radiance {float(b.mean()):.4f}; "
                                    + ("tile split: expected 0" if tile_split else
                                       "sample split sums in another order: expected ~1e-7 relative"))
            # restore this rank's configuration
            tr.SetAccumMode(pt.ACCUM_RUNNING_MEAN if tile_split else pt.ACCUM_SUM)
            tr.SetRowBands(16, rank, world) if tile_split else tr.SetRowBands(16, 0, 1)

    # Average launch duration of the render kernel: at N=1 the timed region is nothing but the
    # K launches, so the event pair around it gives it directly; with a reduce in the step the
    # library's own event pair around the last launch is used instead.
    launch_ms = region_ms / args.steps if not grouped else grouped_launch_ms
    rays_per_launch = rays_local / max(1, args.steps)
    fpr = flops_per_ray_bvh(args.workload) if bvh else flops_per_ray(len(sph), 0 if tris is None else len(tris))
    # HBM bytes per step from the committed PMC passes of this same command (rocprofv3 cannot be
    # run from inside the benchmark): profiles/pmc_traffic_<workload>_v<variant>.json
    launch = tr.LastLaunch()
    global LAUNCH_NAMES
    LAUNCH_NAMES = {
        pt.LAUNCH_MEGAKERNEL: ("v0", "render_kernel"),
        pt.LAUNCH_MEGAKERNEL_STEALING: ("v500", "render_steal_kernel"),
        pt.LAUNCH_WAVEFRONT: ("v100", "wf_step_* (all launches of a step)"),
        pt.LAUNCH_WAVEFRONT_WHOLE_PATHS: ("v100whole", "wf_step_bvh_kernel<whole paths> (all launches of a step)"),
        pt.LAUNCH_PATH_POOL: ("pool", "wf_pool_kernel (+ wf_resolve_kernel; all launches of a step)"),
    }
    launch_tag, launch_kernel = LAUNCH_NAMES.get(launch, ("none", "none"))
    tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.workload}{'_bvh' if bvh else ''}_{launch_tag}.json")
    build_id = pt.build_id()
    pmc, counters_from = pmc_summary(tpath, build_id) if args.integrator == "path" and not args.serial_launches else \
        ({}, {"file": None, "note": "no committed PMC pass for this command"})
    traffic = pmc.get("hbm_bytes_per_step")
    achieved = rays_per_launch * fpr / (launch_ms * 1e-3) / 1e12 if fpr else None

    if rank == 0:
        out = {
            "metric": "path-bounces/sec",
            "value": rays_total / dt_max,
            "unit": "path-bounces/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "build_id": build_id,
            "config": {"workload": wl["label"], "serial_launches": bool(args.serial_launches), "paths_per_step_per_gpu": wl["width"] * wl["height"] * 4 * n_it // (world if tile_split else 1),
                       "bounces_per_step_per_gpu": rays_per_launch, "spheres": int(len(sph)),
                       "triangles": 0 if tris is None else int(len(tris)),
                       "parallelism": "single" if world == 1 else
                                      f"{'tile (16-row bands)' if tile_split else 'sample-range'} split x{world} + "
                                      f"{'RCCL' if backend == 'nccl' else backend + ' (rehearsal, staged through host memory)'} "
                                      f"reduce(sum)" + (" overlapped with the next render" if backend == "nccl" else ""),
                       "kernel_variant": args.variant, "launch": launch_kernel,
                       "nearest_hit": "sphere BVH" if bvh else "linear scan", "integrator": args.integrator},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS if achieved else None,
                         # the peak counts a fused multiply-add as 2 flops; the sphere test may not
                         # contract (bit-exact parity), so half of it is reachable with plain ops
                         "frac_of_unfused_peak": achieved / (PEAK_FP32_VALU_TFLOPS / 2) if achieved else None,
                         "traffic": traffic,
                         # from the committed PMC passes of this command (profiles/): share of the SIMDs'
                         # VALU issue slots in use, lanes active per VALU instruction, wave time in s_waitcnt
                         "valu_issue_util": pmc.get("valu_issue_util"),
                         "active_lane_frac": pmc.get("active_lane_frac"),
                         "wait_frac": pmc.get("wait_frac"),
                         "counters_from": counters_from,
                         "kernel": launch_kernel,
                         "launch_ms": launch_ms,
                         "flops_per_bounce": fpr, "bounces_per_launch": rays_per_launch},
        }
        if multi is not None:
            out["multi_gpu"] = multi
        if (not grouped and not args.no_other_workloads and args.workload == "demo1080" and args.variant == -1
                and args.accel == "auto" and args.integrator == "path"):
            # The other benchmark configurations, a few steps each with the defaults, so that one
            # run of this file measures all of them (not part of `value`, outside its timed region).
            others = []
            for name, accel, k in (("mixed16", "auto", 5), ("synth10k", "auto", 4), ("synth10k", "brute", 3),
                                   ("demo4k", "auto", 3), ("triroom1080", "auto", 3)):
                try:
                    others.append(measure_other_workload(pt, torch, name, accel, k))
                except Exception as e:   # the headline line must not depend on these
                    others.append({"workload": WORKLOADS[name]["label"], "error": f"{type(e).__name__}: {e}"})
            out["other_workloads"] = others
        if not grouped and not args.no_cpu_baseline and args.integrator == "path":
            def gpu_render(n):
                tr.Reset()
                tr.RunN(buf, n, sync=True)
                return buf.cpu().numpy().reshape(wl["height"], wl["width"], 3)
            out["cpu_baseline"] = cpu_baseline(pt, wl, mats, sph, lights, cam, gpu_render, tris)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    tr.Release()
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
