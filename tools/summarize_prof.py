#!/usr/bin/env python3
"""Condense gpurun_out/prof/<tag>/ (rocprofv3 csv output of tools/profile_box.sh) into one
markdown summary for profiles/: kernel-trace stats of the default (overlapped) and of the serial
(one stream) run, launch metadata per kernel, per-step PMC counters and the derived VALU
utilisation / active-lane / occupancy figures of the render kernels.

    python tools/summarize_prof.py <tag> [profiles/pmc_traffic_<...>.json] > profiles/<tag>.md

The optional JSON is what bench.py quotes in `roofline` (HBM bytes, issue-slot utilisation,
active lanes, wait fraction); it carries the build id of the library the counters came from.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RENDER_KERNELS = ("render_", "wf_")
# fields of the bench line that bench.py copies from a committed counter file: not measured by the run shown
REPLAYED = ("traffic", "valu_issue_util", "active_lane_frac", "wait_frac", "counters_from")


def bench_line(log_path):
    if not os.path.exists(log_path):
        return None
    for line in open(log_path).read().splitlines():
        if line.startswith("{"):
            return json.loads(line)
    return None


def short(name, n=70):
    return name if len(name) <= n else name[:n - 3] + "..."


def kernel_stats(path, title, n_steps):
    """Prints kt_kernel_stats.csv; returns {kernel name: total ns} of the render kernels."""
    print(f"## {title}\n")
    with open(path) as f:
        rows = list(csv.DictReader(f))
    cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
    print("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols))
    totals = {}
    for r in rows:
        print("| " + " | ".join([short(r["Name"])] + [r[c] for c in cols[1:]]) + " |")
        if any(k in r["Name"] for k in RENDER_KERNELS):
            totals[r["Name"]] = float(r["TotalDurationNs"])
    print()
    return totals


def compiler_resources():
    """{kernel: (VGPRs, SGPRs, waves per SIMD)} from the newest profiles/*_kernel_resources.md
    (tools/kernel_resources.py: the compiler's own figures)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernel_resources.md")))
    out = {}
    if files:
        for line in open(files[-1]):
            f = [x.strip() for x in line.strip().strip("|").split("|")]
            if len(f) == 7 and f[0].startswith("`"):
                out[f[0].strip("`")] = (f[1], f[3], f[6])
    return out, (os.path.relpath(files[-1], ROOT) if files else None)


def main():
    tag = sys.argv[1]
    base = os.path.join(ROOT, "gpurun_out", "prof", tag)
    print(f"# rocprofv3 summary `{tag}`\n")
    j = bench_line(os.path.join(base, "kt.log"))
    n_steps = j["steps"] + j["warmup"] if j else 1
    if j:
        shown = json.loads(json.dumps(j))
        for k in REPLAYED:
            shown.get("roofline", {}).pop(k, None)
        print(f"library build id: `{j.get('build_id')}`\n")
        print("bench line under the profiler (the counter fields that bench.py copies from a committed file are "
              "left out: this run did not measure them):\n\n```json\n" + json.dumps(shown) + "\n```\n")

    kernel_stats(os.path.join(base, "kt", "kt_kernel_stats.csv"),
                 "`rocprofv3 --kernel-trace --stats`, default run (batches overlap on two streams: durations are NOT additive)", n_steps)
    serial_csv = os.path.join(base, "kt_serial", "kt_kernel_stats.csv")
    if os.path.exists(serial_csv):
        js = bench_line(os.path.join(base, "kt_serial.log"))
        ns = js["steps"] + js["warmup"] if js else n_steps
        totals = kernel_stats(serial_csv, "`rocprofv3 --kernel-trace --stats`, serial run (`bench.py --serial-launches`: one buffer set, "
                              "one stream, every launch after the one before)", ns)
        if js and totals:
            step_ms = sum(totals.values()) / ns / 1e6
            dom = max(totals, key=totals.get)
            dom_ms = totals[dom] / ns / 1e6
            r = js["roofline"]
            flops = r["bounces_per_launch"] * r["flops_per_bounce"] if r.get("flops_per_bounce") else None
            print(f"* render kernels per step, serial run: {step_ms:.3f} ms (sum of their durations / {ns} steps); "
                  f"the run's own clock: {js['ms_per_step']:.3f} ms per step")
            print(f"* dominant kernel `{short(dom, 60)}`: {dom_ms:.3f} ms per step")
            if flops:
                print(f"* roofline from this trace: {r['bounces_per_launch']:.0f} bounces x {r['flops_per_bounce']} flop / "
                      f"{step_ms:.3f} ms = {flops / step_ms / 1e9:.2f} TFLOP/s = {flops / step_ms / 1e9 / r['peak']:.4f} of {r['peak']} "
                      f"(dominant kernel alone: {flops / dom_ms / 1e9 / r['peak']:.4f})")
            print()

    # PMC: per kernel launch metadata, and counters summed over the render kernels per bench step
    sums, counts = defaultdict(float), defaultdict(int)
    per_kernel = defaultdict(lambda: defaultdict(float))
    meta = {}
    for path in sorted(glob.glob(os.path.join(base, "*", "pmc_counter_collection.csv"))):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                if not any(k in name for k in RENDER_KERNELS):
                    continue
                sums[r["Counter_Name"]] += float(r["Counter_Value"])
                counts[r["Counter_Name"]] += 1
                per_kernel[name][r["Counter_Name"]] += float(r["Counter_Value"])
                m = meta.setdefault(name, {"grids": set()})
                for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size"):
                    if k in r:
                        m[k] = r[k]
                if "Grid_Size" in r:
                    m["grids"].add(int(r["Grid_Size"]))
    if meta:
        comp, comp_file = compiler_resources()
        print("## launch metadata per kernel, largest share of the VALU instructions first\n")
        print("The rocprofv3 columns are what the counter passes' CSV holds for the dispatch: its VGPR_Count is half of "
              "what the compiler allocates for every kernel here (a granule of 4 applied to a field that counts blocks of 8), "
              "and its LDS size leaves out the dynamic LDS the launch asks for.  The compiler's figures"
              + (f" (`{comp_file}`)" if comp_file else "") + " are the ones to read occupancy from.\n")
        print("| kernel | compiler: VGPRs | SGPRs | waves per SIMD | rocprofv3: VGPR_Count | Accum_VGPR_Count | SGPR_Count | LDS_Block_Size | Scratch_Size | workgroup | grid sizes (threads) | share of SQ_INSTS_VALU |")
        print("|" + "---|" * 12)
        total_valu = sum(v.get("SQ_INSTS_VALU", 0.0) for v in per_kernel.values()) or 1.0
        for name in sorted(meta, key=lambda n: -per_kernel[n].get("SQ_INSTS_VALU", 0.0)):
            m = meta[name]
            vg, ag = int(m.get("VGPR_Count", 0)), int(m.get("Accum_VGPR_Count", 0))
            grids = sorted(m["grids"])
            gtxt = ", ".join(map(str, grids[:4])) + (", ..." if len(grids) > 4 else "")
            c = next((v for k, v in comp.items() if k in name), ("?", "?", "?"))
            print(f"| `{short(name, 60)}` | {c[0]} | {c[1]} | {c[2]} | {vg} | {ag} | {m.get('SGPR_Count')} | {m.get('LDS_Block_Size')} | "
                  f"{m.get('Scratch_Size')} | {m.get('Workgroup_Size')} | {gtxt} | {per_kernel[name].get('SQ_INSTS_VALU', 0.0) / total_valu:.3f} |")
        print()
    # the megakernel is one launch per bench step; the wavefront variant and the path pool are
    # several launches per step (bounces or batches, and the resolve), so their counters are summed
    # over all launches of the run and divided by the number of bench steps (warm-up included)
    wavefront = any("wf_" in k for k in meta)
    avg = {k: (sums[k] / n_steps if wavefront else sums[k] / counts[k]) for k in sums}
    print(f"## PMC counters per bench step ({'sum over all wf_* launches of a step' if wavefront else 'one megakernel launch'})\n")
    print("| counter | value |\n|---|---|")
    for k in sorted(avg):
        print(f"| {k} | {avg[k]:.6g} |")
    print()
    g = avg.get
    print("## derived\n")
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
        print(f"* active-lane fraction of VALU work (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)): "
              f"{g('SQ_THREAD_CYCLES_VALU') / (64 * g('SQ_ACTIVE_INST_VALU')):.3f}")
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
        print(f"* VALU instructions per wave: {g('SQ_INSTS_VALU') / g('SQ_WAVES'):.0f}; "
              f"SALU per wave: {g('SQ_INSTS_SALU', 0) / g('SQ_WAVES'):.0f}; LDS per wave: {g('SQ_INSTS_LDS', 0) / g('SQ_WAVES'):.0f}")
    if g("SQ_INSTS_VALU") and j and j.get("roofline", {}).get("bounces_per_launch"):
        b = j["roofline"]["bounces_per_launch"]
        lanes = g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")) if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU") else None
        print(f"* per bounce: {g('SQ_INSTS_VALU') * 64 / b:.0f} VALU lane-slots issued"
              + (f", {g('SQ_INSTS_VALU') * 64 * lanes / b:.0f} with an active lane" if lanes else "")
              + f", {g('SQ_INSTS_SALU', 0) * 64 / b:.0f} SALU x 64, against {j['roofline'].get('flops_per_bounce')} algorithmic flops")
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_BUSY_CYCLES"):
        # SQ_* cycle counters are summed over SEs/XCDs in quad-cycle units; the ratio is what matters
        print(f"* SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES: {g('SQ_ACTIVE_INST_VALU') / g('SQ_BUSY_CYCLES'):.3f}")
    if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
        print(f"* mean resident waves (SQ_WAVE_CYCLES / SQ_BUSY_CYCLES): {g('SQ_WAVE_CYCLES') / g('SQ_BUSY_CYCLES'):.2f}")
    if g("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
            if g(k):
                print(f"* {k} / SQ_WAVE_CYCLES: {g(k) / g('SQ_WAVE_CYCLES'):.3f}")
    if g("GRBM_GUI_ACTIVE") and g("SQ_INSTS_VALU"):
        print(f"* VALU wave-instructions per GRBM_GUI_ACTIVE cycle (sum over XCDs): {g('SQ_INSTS_VALU') / g('GRBM_GUI_ACTIVE'):.2f}")
    if g("FETCH_SIZE") is not None and len(sys.argv) > 2:
        # machine-readable HBM traffic per bench step for bench.py's roofline.traffic:
        # FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B), WRITE_SIZE taken as is;
        # both are reported in KiB
        with open(sys.argv[2], "w") as f:
            extra = {}
            # share of the VALU issue slots in use: a SIMD issues one fp32 wave64 instruction per 2
            # cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs, the chip has 1024 SIMDs
            if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
                extra["valu_issue_util"] = g("SQ_INSTS_VALU") / (1024 * (g("GRBM_GUI_ACTIVE") / 8) / 2)
            if g("SQ_ACTIVE_INST_VALU") and g("SQ_THREAD_CYCLES_VALU"):
                extra["active_lane_frac"] = g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU"))
            if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES"):
                extra["wait_frac"] = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")
            if g("SQ_INSTS_VALU"):
                extra["valu_wave_instructions_per_step"] = g("SQ_INSTS_VALU")
            json.dump({"tag": tag, "build_id": j.get("build_id") if j else None,
                       "fetch_bytes": g("FETCH_SIZE") * 1024 * 2, "write_bytes": g("WRITE_SIZE", 0.0) * 1024,
                       "hbm_bytes_per_step": g("FETCH_SIZE") * 1024 * 2 + g("WRITE_SIZE", 0.0) * 1024, **extra,
                       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, per bench step; "
                               "FETCH_SIZE x2 (gfx950 correction for wide coalesced reads); build_id = first 16 hex "
                               "digits of the SHA-256 of the libptmi.so the counters were taken from"}, f, indent=1)
    if g("FETCH_SIZE") is not None:
        print(f"* FETCH_SIZE {g('FETCH_SIZE'):.6g} KB as reported (x2 on gfx950 for 16 B/lane streaming reads), "
              f"WRITE_SIZE {g('WRITE_SIZE', float('nan')):.6g} KB")


if __name__ == "__main__":
    main()
