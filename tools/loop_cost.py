#!/usr/bin/env python3
"""Cost-weighted account of k_sample<2>'s pool loop from the compiler's ISA.

usage: make -C vcfgl_amd/csrc asm && python tools/loop_cost.py [mangled-kernel-substring] > profiles/r04_loop_cost.json

Finds the innermost pool loop (the one that holds v_rcp_f64) of the chosen instantiation (default: the C3 build
k_sample<2, false, 1, false, 2>), counts the vector instructions of each basic block and prices them with the issue costs
measured on this part by tools/valu_rates.hip (cycles of SIMD time per wavefront instruction, one wave's stream):
  float64 fma / mul / add, v_mul_lo/hi_u32, v_mad_u64_u32, 64-bit shifts and compares, three-operand VOP3 forms   4.3
  two-operand 32-bit VOP2 forms, v_fma_f32 / v_fmac_f32, v_cndmask, v_mov                                          2.6
  v_log_f32 / v_rcp_f32 / v_exp_f32                                                                               8.3
  v_rcp_f64 / v_sqrt_f64                                                                                         16.3
Blocks: `common` = executed by every iteration; `finish` = the block a lane that finishes a read runs (nearly every iteration has
one); `slow_n` / `slow_g` = the bounded-log tests, entered every P.slow_period-th iteration when a lane asks (weights below)."""
import json
import re
import sys

ASM = "build/asm/vgl_sample-hip-amdgcn-amd-amdhsa-gfx950.s"
KEY = sys.argv[1] if len(sys.argv) > 1 else "_Z8k_sampleILi2ELb0ELi1ELb0ELi2EEv"

COST = {"f64": 4.3, "int64": 4.3, "vop3": 4.3, "vop2": 2.6, "trans32": 8.3, "trans64": 16.3}
THREE_OP = ("v_add3_u32", "v_lshl_add_u32", "v_and_or_b32", "v_lshl_or_b32", "v_mad_u32_u24", "v_fma_f32", "v_bfe_u32", "v_alignbit_b32", "v_perm_b32", "v_lshl_add_u64")


def klass(op):
    if op in ("v_rcp_f64_e32", "v_sqrt_f64_e32", "v_rsq_f64_e32"):
        return "trans64"
    if re.match(r"v_(log|exp|rcp|rsq|sqrt|sin|cos)_f32", op):
        return "trans32"
    if "_f64" in op and not op.startswith("v_cvt_f32"):
        return "f64"
    if op.startswith(("v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshlrev_b64", "v_lshrrev_b64", "v_cmp_gt_u64", "v_cmp_lt_u64", "v_cmp_gt_i64", "v_mul_u64")):
        return "int64"
    if op.startswith("v_fma_f32") or op.startswith("v_fmac_f32") or op.startswith("v_fmamk_f32") or op.startswith("v_pk_"):
        return "vop2"                                         # measured with the two-operand class
    if op.startswith(THREE_OP) or op.endswith("_e64"):
        return "vop3"
    return "vop2"


def main():
    text = open(ASM).read().split("\n")
    start = next(i for i, l in enumerate(text) if l.startswith(KEY) and l.rstrip().endswith(tuple(": ;@")) or (l.startswith(KEY) and ":" in l))
    end = next(i for i in range(start, len(text)) if text[i].startswith(".Lfunc_end"))
    body = text[start:end]
    rcp = next(i for i, l in enumerate(body) if "v_rcp_f64" in l)
    hdr = max(i for i in range(rcp) if "Inner Loop Header" in body[i]) - 1
    latch = max(i for i in range(hdr) if body[i].startswith(".LBB"))
    latch_label = body[latch].split(":")[0]
    last = min(i for i in range(rcp, len(body)) if ("s_branch " + latch_label) in body[i])
    blocks, cur = [], None
    for l in body[latch:last + 1]:
        if l.startswith(".LBB") or l.startswith("; %bb"):
            cur = {"label": l.split(":")[0].strip("; "), "valu": 0, "salu": 0, "mem": 0, "cycles": 0.0, "classes": {}}
            blocks.append(cur)
            continue
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        op = t.split()[0]
        if op.startswith("v_"):
            k = klass(op)
            cur["valu"] += 1; cur["cycles"] += COST[k]; cur["classes"][k] = cur["classes"].get(k, 0) + 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith(("ds_", "global_", "buffer_", "flat_")):
            cur["mem"] += 1
    # roles: the two big blocks of the bounded tests hold v_log_f32; the finishing block holds the LDS atomic / ds_write
    roles = {}
    for b in blocks:
        txt = "\n".join(body[latch:last + 1])
    for b in blocks:
        seg = []
        on = False
        for l in body[latch:last + 1]:
            if l.startswith(".LBB") or l.startswith("; %bb"):
                on = l.split(":")[0].strip("; ") == b["label"]
                continue
            if on:
                seg.append(l)
        s = "\n".join(seg)
        if "v_log_f32" in s:
            b["role"] = "slow_n" if "v_ldexp_f64" in s else "slow_g"
        elif "ds_add_rtn" in s or "ds_write_b32" in s:
            b["role"] = "finish"
        elif b["valu"] <= 2 and b["salu"] <= 4:
            b["role"] = "slow_gate"
        else:
            b["role"] = "common"
    common = [b for b in blocks if b["role"] == "common"]
    finish = [b for b in blocks if b["role"] == "finish"]
    slow = [b for b in blocks if b["role"] in ("slow_n", "slow_g", "slow_gate")]
    # weights: the finishing block runs whenever ANY lane finishes a read (each lane does every ~3.3 iterations: in practice every
    # iteration); the bounded tests every 4th iteration (P.slow_period), entered when a lane has asked since: ~0.9 (normal) / ~0.4 (gamma)
    w_slow = {"slow_n": 0.9 / 4, "slow_g": 0.4 / 4, "slow_gate": 0.25}
    per_iter_valu = sum(b["valu"] for b in common) + sum(b["valu"] for b in finish) + sum(b["valu"] * w_slow[b["role"]] for b in slow)
    per_iter_cyc = sum(b["cycles"] for b in common) + sum(b["cycles"] for b in finish) + sum(b["cycles"] * w_slow[b["role"]] for b in slow)
    out = {"kernel": KEY, "blocks": blocks,
           "valu_common_path": sum(b["valu"] for b in common), "valu_finish_block": sum(b["valu"] for b in finish),
           "valu_per_iteration_weighted": round(per_iter_valu, 1), "simd_cycles_per_iteration_weighted": round(per_iter_cyc, 1),
           "avg_cycles_per_valu_inst": round(per_iter_cyc / per_iter_valu, 3),
           "cost_table_cycles": COST, "cost_source": "tools/valu_rates.hip (MI355X, one wave's stream; DESIGN.md section 5)"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
