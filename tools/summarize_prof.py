#!/usr/bin/env python3
"""Condense gpurun_out/prof/<tag>/ (rocprofv3 csv output of tools/profile_box.sh) into one
markdown summary for profiles/: kernel-trace stats plus per-launch PMC counters and the
derived VALU utilisation / active-lane / occupancy figures of the render kernel.

    python tools/summarize_prof.py <tag> > profiles/<tag>.md
"""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    base = os.path.join(ROOT, "gpurun_out", "prof", tag)
    print(f"# rocprofv3 summary `{tag}`\n")
    cmd = open(os.path.join(base, "kt.log")).read()
    for line in cmd.splitlines():
        if line.startswith("{"):
            print("bench line under the profiler:\n\n```json\n" + line + "\n```\n")
    print("## `rocprofv3 --kernel-trace --stats` (kt_kernel_stats.csv)\n")
    with open(os.path.join(base, "kt", "kt_kernel_stats.csv")) as f:
        rows = list(csv.DictReader(f))
    cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
    print("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols))
    for r in rows:
        name = r["Name"]
        if len(name) > 70:
            name = name[:67] + "..."
        print("| " + " | ".join([name] + [r[c] for c in cols[1:]]) + " |")
    print()

    # PMC: average per launch of the render kernel
    sums, counts = defaultdict(float), defaultdict(int)
    meta = {}
    kernels = set()
    for path in sorted(glob.glob(os.path.join(base, "*", "pmc_counter_collection.csv"))):
        with open(path) as f:
            for r in csv.DictReader(f):
                if "render_" not in r["Kernel_Name"] and "wf_" not in r["Kernel_Name"]:
                    continue
                sums[r["Counter_Name"]] += float(r["Counter_Value"])
                counts[r["Counter_Name"]] += 1
                kernels.add(r["Kernel_Name"])
                meta = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                          "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r}
    # the megakernel is one launch per bench step; the wavefront variant and the path pool are
    # several launches per step (bounces or batches, and the resolve), so their counters are summed
    # over all launches of the run and divided by the number of bench steps (warm-up included) to
    # stay comparable
    n_steps = 1
    for line in cmd.splitlines():
        if line.startswith("{"):
            import json
            j = json.loads(line)
            n_steps = j["steps"] + j["warmup"]
    wavefront = any("wf_" in k for k in kernels)
    avg = {k: (sums[k] / n_steps if wavefront else sums[k] / counts[k]) for k in sums}
    print(f"## PMC counters per bench step ({'sum over all wf_* launches of a step' if wavefront else 'one megakernel launch'})\n")
    print("launch: " + ", ".join(f"{k}={v}" for k, v in meta.items()) + "\n")
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
        import json
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
            json.dump({"tag": tag, "fetch_bytes": g("FETCH_SIZE") * 1024 * 2, "write_bytes": g("WRITE_SIZE", 0.0) * 1024,
                       "hbm_bytes_per_step": g("FETCH_SIZE") * 1024 * 2 + g("WRITE_SIZE", 0.0) * 1024, **extra,
                       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, per bench step; "
                               "FETCH_SIZE x2 (gfx950 correction for wide coalesced reads)"}, f, indent=1)
    if g("FETCH_SIZE") is not None:
        print(f"* FETCH_SIZE {g('FETCH_SIZE'):.6g} KB as reported (x2 on gfx950 for 16 B/lane streaming reads), "
              f"WRITE_SIZE {g('WRITE_SIZE', float('nan')):.6g} KB")


if __name__ == "__main__":
    main()
