#!/usr/bin/env python3
"""Static instruction budget of one kernel of pt_wavefront.hip by source function and issue class.

Compiles the file to gfx950 assembly with line tables, takes the body of the kernel whose mangled name
contains <kernel substring>, attributes every instruction to the source function its line belongs to
(inlined code keeps its own lines) and sorts the mnemonics into issue classes with the costs measured by
tools/microbench/valu_rates.hip (cycles per wave64 instruction at full occupancy):

    full   2   v_add/sub/mul/fma/fmac/mac_f32, logic, shifts, integer add, v_mov, v_alignbit, v_perm
    half   4   v_cmp*, v_cndmask, v_min/max/med3, conversions, v_mul_lo/hi, v_mad_u32_u24, readlane ...
    trans  8   v_rcp/rsq/sqrt/sin/cos/exp/log_f32
    mad64  8   v_mad_u64_u32 (Philox)

The counts are static: code inside a loop or behind a branch counts once however often it runs.  For the
path pool on a scene of up to 16 spheres the pass is straight-line code apart from the blocks named in
the output, so the static figures of the scan (per sphere x spheres), the Philox block and the BSDF are
what one pass executes.

    python tools/instruction_budget.py [kernel substring] > profiles/r3_pool_instruction_budget.md"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cuda-pathtracer_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Iinclude", "-I" + CSRC, "-gline-tables-only", "-S", "--offload-device-only"]

FULL = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad)_f32|^v_(and|or|xor|not|lshl|lshr|ashr|bfe|bfi|and_or|or3|xad|lshl_add|lshl_or|add_lshl|add3|add|sub|subrev|addc|subb)_|^v_mov_b32|^v_alignbit|^v_perm|^v_bitop|^v_xor3|^v_lshlrev|^v_lshrrev|^v_ashrrev|^v_accvgpr|^v_swap|^v_nop|^v_pk_")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|sin|cos|exp|log)_")
MAD64 = re.compile(r"^v_mad_[ui]64_")


def issue_class(m):
    if not m.startswith("v_"):
        return "lds" if m.startswith("ds_") else "vmem" if m.startswith(("global_", "buffer_", "flat_", "scratch_")) else \
            "branch" if m.startswith(("s_cbranch", "s_branch")) else "wait" if m.startswith(("s_waitcnt", "s_nop", "s_barrier")) else "salu"
    if TRANS.match(m):
        return "trans"
    if MAD64.match(m):
        return "mad64"
    if FULL.match(m):
        return "full"
    return "half"


def function_ranges(path):
    """[(first line, name)] of the function definitions of a source file (device functions and kernels)."""
    out = []
    pat = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:PT_DEV|__global__|static|inline)\b[^;=]*?\b([A-Za-z_][A-Za-z_0-9]*)\s*\(")
    with open(path) as f:
        for no, line in enumerate(f, 1):
            m = pat.match(line)
            if m and m.group(1) not in ("if", "for", "while", "switch", "return", "__launch_bounds__"):
                out.append((no, m.group(1)))
            elif "__launch_bounds__" in line and "void" in line:
                m2 = re.search(r"void\s+([A-Za-z_0-9]+)\s*\(", line)
                if m2:
                    out.append((no, m2.group(1)))
    return out


def main():
    sub = sys.argv[1] if len(sys.argv) > 1 else "wf_pool_kernelILb0ELi0ELb0E"
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "wf.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", asm, os.path.join(CSRC, "pt_wavefront.hip")], check=True, cwd=ROOT,
                       stderr=subprocess.DEVNULL)
        lines = open(asm).read().splitlines()
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.join(m.group(2), m.group(3)) if m.group(3) else m.group(2)
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(sub) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ranges = {}
    counts = collections.defaultdict(collections.Counter)
    cur = ("?", 0)
    for l in lines[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if not m or m.group(1).startswith("."):
            continue
        path, no = cur
        base = os.path.basename(path)
        if path not in ranges:
            full = path if os.path.isabs(path) else os.path.join(ROOT, path)
            ranges[path] = function_ranges(full) if os.path.exists(full) and base.startswith("pt_") else []
        fn = "(other)"
        for first, name in ranges[path]:
            if first <= no:
                fn = name
            else:
                break
        counts[(base, fn)][issue_class(m.group(1))] += 1
    cost = {"full": 2, "half": 4, "trans": 8, "mad64": 8}
    print(f"# static instruction budget of `{sub}` (gfx950, the library's flags)\n")
    print((__doc__ or "").split("\n\n")[1].replace("\n", " ") + "\n")
    print("| source function | full-rate VALU | half-rate VALU | transcendental | v_mad_u64_u32 | VALU issue cycles | SALU | branches | LDS | vector memory |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    total = collections.Counter()
    rows = sorted(counts.items(), key=lambda kv: -sum(kv[1][c] * cost[c] for c in cost))
    for (base, fn), c in rows:
        cyc = sum(c[k] * cost[k] for k in cost)
        total.update(c)
        print(f"| `{fn}` ({base}) | {c['full']} | {c['half']} | {c['trans']} | {c['mad64']} | {cyc} | {c['salu']} | {c['branch']} | {c['lds']} | {c['vmem']} |")
    copies = sum(c["lds"] for (base, fn), c in counts.items() if fn == "scan_lds")
    if copies:
        scan = collections.Counter()
        for (base, fn), c in counts.items():
            if fn in ("sphere_test", "sphere_accept", "pt_sqrt_discriminant", "scan_lds"):
                scan.update(c)
        newton = counts.get(("pt_device.hpp", "sqrt_newton"), collections.Counter())
        print(f"\nThe scan is present {copies} times in the code (one `ds_read_b128` each: unrolled groups for scenes of up to 16 "
              f"spheres, the loop for larger ones and its tail).  Per copy: {scan['full'] / copies:.1f} full-rate + "
              f"{scan['half'] / copies:.1f} half-rate + {scan['trans'] / copies:.1f} transcendental instructions in the test "
              f"itself (the transcendental and part of the half-rate ones belong to the generic square root behind the range "
              f"gate, which a discriminant above 2^-100 does not take), plus the short square root: "
              f"{newton['full'] / max(newton['trans'], 1):.1f} full-rate + 1 `v_rsq_f32`.\n")
        print("| source function | full-rate VALU | half-rate VALU | transcendental | v_mad_u64_u32 | VALU issue cycles | SALU | branches | LDS | vector memory |")
        print("|---|---|---|---|---|---|---|---|---|---|")
    cyc = sum(total[k] * cost[k] for k in cost)
    print(f"| **whole kernel (static)** | {total['full']} | {total['half']} | {total['trans']} | {total['mad64']} | {cyc} | {total['salu']} | {total['branch']} | {total['lds']} | {total['vmem']} |")


if __name__ == "__main__":
    main()
