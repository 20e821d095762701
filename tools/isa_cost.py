#!/usr/bin/env python3
"""Static issue-cost estimate of a kernel's basic blocks from hipcc's .s output, using the
per-instruction issue costs measured by tools/microbench/valu_rates.hip on gfx950 (cycles per
wave64 instruction at full occupancy): fp32 add/mul/fma, logic, mov, shifts, u32 add: 2.2;
the same with an SGPR operand, compares, cndmask, min/max, cvt, integer multiply, div helpers:
4; v_pk_*: 4.3; rcp/rsq/sqrt: 7.7.  Prints one line per label: VALU instructions and cycles.

    tools/isa_cost.py file.s kernel_symbol_substring
"""
import re
import sys

FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_xor_b32",
        "v_and_b32", "v_or_b32", "v_mov_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_ashrrev_i32", "v_not_b32", "v_mov_b64", "v_mul_legacy_f32")
TRANS = ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32")


def cost(line):
    op = line.split()[0]
    base = op.replace("_e32", "").replace("_e64", "").replace("_dpp", "").replace("_sdwa", "")
    if not base.startswith("v_"):
        return None
    if base in TRANS:
        return 7.7
    if base.startswith("v_pk_"):
        return 4.3
    if base in FAST:
        operands = line.split(None, 1)[1] if len(line.split(None, 1)) > 1 else ""
        ops = [o.strip() for o in operands.split(",")][1:]
        if any(re.match(r"^-?\|?s(\d+|\[)", o) or o in ("vcc_lo", "vcc_hi") or re.match(r"^0x[0-9a-f]+$", o) for o in ops):
            return 4.0
        return 2.2
    return 4.0


def main():
    path, sym = sys.argv[1], sys.argv[2]
    active = False
    label, n, cyc, salu, lds = "entry", 0, 0.0, 0, 0
    tot_n = tot_c = 0
    for raw in open(path):
        line = raw.strip()
        if not active:
            if line.startswith("_Z") and sym in line and line.endswith(":") or (sym in line and re.match(r"^_Z\S+:", line)):
                active = True
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            print(f"{label:12s} valu {n:4d}  cycles {cyc:7.1f}  salu {salu:3d} lds {lds:2d}")
            label, n, cyc, salu, lds = m.group(1), 0, 0.0, 0, 0
            continue
        if not line or line.startswith(";") or line.startswith("."):
            continue
        c = cost(line)
        if c is not None:
            n += 1; cyc += c; tot_n += 1; tot_c += c
        elif line.startswith("s_"):
            salu += 1
        elif line.startswith("ds_"):
            lds += 1
    print(f"{label:12s} valu {n:4d}  cycles {cyc:7.1f}  salu {salu:3d} lds {lds:2d}")
    print(f"total valu {tot_n} cycles {tot_c:.0f}")


if __name__ == "__main__":
    main()
