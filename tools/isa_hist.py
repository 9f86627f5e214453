#!/usr/bin/env python3
"""tools/isa_hist.py -- cycle-weighted issue cost of a kernel's hot loop, from the compiler's own assembly.

What it answers: how many cycles of a SIMD's vector pipe one iteration of a loop needs when nothing
stalls, so that "this kernel is at x of its instruction-issue roof" can be reproduced from files under profiles/:

    issue cycles per iteration = sum over the loop's blocks of  weight(block) x sum over its vector instructions of  cycles(mnemonic)

* the instructions come from build/asm/<file>-hip-amdgcn-amd-amdhsa-gfx950.s (`make -C vcfgl_amd/csrc asm`),
* cycles(mnemonic) from the output of tools/valu_rates.hip on the GPU box (profiles/r06_valu_rates.txt: SIMD cycles per
  wavefront instruction with four wavefronts per SIMD, i.e. issue throughput, not latency),
* weight(block) = how often the block runs per iteration.  Blocks on the loop's every-iteration path weigh 1; the weight of a
  block behind a wave-uniform or a divergent branch is given on the command line (--weight LABEL=w, with the reason in --note)
  and is printed with the result -- nothing is guessed silently.

The roof the bench line uses (bench.py `roofline.issue_*`):
    issue_cycles_per_inst = (weighted cycles of the loop) / (weighted vector instructions of the loop)
    issue_bound_ms        = waves per launch x VALU instructions per wave (SQ_INSTS_VALU / SQ_WAVES of the committed counters)
                            x issue_cycles_per_inst / (1024 SIMDs x sclk)
    issue_frac            = issue_bound_ms / measured launch time          (<= 1 up to the error of the mix)
i.e. the loop's instruction MIX prices every dynamic vector instruction of the kernel (the loop is where > 80 % of them are).

usage: python tools/isa_hist.py --asm build/asm/vgl_sample-hip-amdgcn-amd-amdhsa-gfx950.s --kernel 'k_sample<2, false, 1, false, 2>' \
           --loop-with ds_add_rtn_u32 --rates profiles/r06_valu_rates.txt [--weight .LBB14_93=0.1 ...] [--json out.json] [--list]
"""
import argparse
import json
import re
import subprocess
import sys

# valu_rates kernel name -> the mnemonics it prices (without _e32 / _e64 / _sdwa / _dpp suffixes)
RATE_KERNELS = {
    "k_fma_f64": ["v_fma_f64"], "k_mul_f64": ["v_mul_f64"], "k_add_f64": ["v_add_f64"], "k_rcp_f64": ["v_rcp_f64", "v_rsq_f64"], "k_sqrt_f64": ["v_sqrt_f64"],
    "k_ldexp_f64": ["v_ldexp_f64"], "k_lshl_b64": ["v_lshlrev_b64"], "k_lshr_b64": ["v_lshrrev_b64", "v_ashrrev_i64"],
    "k_cmp_f64": ["v_cmp_*_f64", "v_cmp_class_f64"], "k_max_f64": ["v_max_f64", "v_min_f64"],
    "k_fract_f64": ["v_fract_f64"], "k_floor_f64": ["v_floor_f64", "v_trunc_f64", "v_rndne_f64", "v_ceil_f64"],
    "k_cvt_f32_f64": ["v_cvt_f32_f64"], "k_cvt_f64_f32": ["v_cvt_f64_f32"], "k_cvt_f64_u32_real": ["v_cvt_f64_u32"], "k_cvt_f64_i32": ["v_cvt_f64_i32"],
    "k_cvt_i32_f64": ["v_cvt_i32_f64", "v_cvt_u32_f64"],
    "k_mul_lo_u32": ["v_mul_lo_u32"], "k_mul_hi_u32": ["v_mul_hi_u32"], "k_mul_hi_i32": ["v_mul_hi_i32"], "k_mul_u24": ["v_mul_u32_u24", "v_mul_i32_i24"],
    "k_mul_hi_u24": ["v_mul_hi_u32_u24"], "k_mad_u24": ["v_mad_u32_u24", "v_mad_i32_i24"], "k_mad_u64_s": ["v_mad_u64_u32", "v_mad_i64_i32"],
    "k_add_u32": ["v_add_u32", "v_add_nc_u32"], "k_sub_u32": ["v_sub_u32"], "k_subrev_u32": ["v_subrev_u32"], "k_add3_u32": ["v_add3_u32"],
    "k_lshl_add": ["v_lshl_add_u32"], "k_add_lshl": ["v_add_lshl_u32"], "k_lshl_or": ["v_lshl_or_b32"], "k_and_or": ["v_and_or_b32", "v_or3_b32", "v_xad_u32"],
    "k_cndmask_s": ["v_cndmask_b32"], "k_cmp_u32_s": ["v_cmp_*_u32", "v_cmp_*_i32", "v_cmp_*_u16", "v_cmp_*_i16", "v_cmp_*_u64", "v_cmp_*_i64"],
    "k_cmp_f32_s": ["v_cmp_*_f32", "v_cmp_class_f32", "v_cmpx_*"],
    "k_mov": ["v_mov_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"], "k_and": ["v_and_b32"], "k_or": ["v_or_b32"], "k_xor_lit": ["v_xor_b32", "v_not_b32"],
    "k_lshl": ["v_lshlrev_b32"], "k_lshr": ["v_lshrrev_b32", "v_ashrrev_i32"], "k_bfe_u32": ["v_bfe_u32", "v_bfe_i32", "v_bfi_b32"],
    "k_mbcnt": ["v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"], "k_bcnt": ["v_bcnt_u32_b32", "v_ffbh_u32", "v_ffbl_b32"],
    "k_cvt_f64_u32": ["v_cvt_f32_u32", "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte1", "v_cvt_f32_ubyte2", "v_cvt_f32_ubyte3"], "k_cvt_f32_i32": ["v_cvt_f32_i32"],
    "k_cvt_i32_f32": ["v_cvt_i32_f32", "v_cvt_flr_i32_f32", "v_cvt_rpi_i32_f32"], "k_cvt_u32_f32": ["v_cvt_u32_f32"],
    "k_log_f32": ["v_log_f32"], "k_rcp_f32": ["v_rcp_f32", "v_rsq_f32", "v_rcp_iflag_f32"], "k_cvt_f64_f32_lo": ["v_exp_f32"], "k_sqrt_f32": ["v_sqrt_f32", "v_sin_f32", "v_cos_f32"],
    "k_fma_f32": ["v_fma_f32", "v_mad_f32"], "k_fmac_f32": ["v_fmac_f32", "v_mac_f32"], "k_fmamk_f32": ["v_fmamk_f32", "v_fmaak_f32", "v_madmk_f32", "v_madak_f32"],
    "k_mul_f32": ["v_mul_f32"], "k_add_f32": ["v_add_f32", "v_sub_f32", "v_subrev_f32"], "k_max_f32": ["v_max_f32", "v_min_f32", "v_max3_f32", "v_min3_f32", "v_med3_f32"],
    "k_min_u32": ["v_min_u32", "v_max_u32", "v_min_i32", "v_max_i32", "v_med3_i32", "v_med3_u32", "v_max3_u32", "v_min3_u32"],
    "k_ldexp_f32": ["v_ldexp_f32", "v_frexp_mant_f32", "v_frexp_exp_i32_f32", "v_fract_f32", "v_floor_f32", "v_trunc_f32", "v_rndne_f32"],
    "k_alignbit": ["v_alignbit_b32", "v_alignbyte_b32"], "k_perm": ["v_perm_b32"],
    "k_pk_mul_f32": ["v_pk_mul_f32"], "k_pk_add_f32": ["v_pk_add_f32"], "k_pk_fma_f32": ["v_pk_fma_f32"],
    "k_addc": ["v_addc_co_u32", "v_subb_co_u32", "v_subbrev_co_u32"], "k_add_co": ["v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32"],
    "k_readlane": ["v_readlane_b32", "v_writelane_b32"], "k_readfirst": ["v_readfirstlane_b32"],
}
# forms priced by their own kernel when the modifier is present
SUFFIX_KERNELS = {"_sdwa": "k_add_sdwa", "_dpp": "k_mov_dpp"}
DEFAULT_CYCLES = 4.3       # the three-operand / 64-bit-encoded class: what an unlisted vector mnemonic is priced at (listed in the output)


def load_rates(path, waves=4):
    """{kernel name: cycles per wavefront instruction (SIMD time)} at `waves` wavefronts per SIMD"""
    r = {}
    for l in open(path):
        m = re.match(r"(\S+)\s+waves/SIMD (\d+): ([\d.]+) cyc", l)
        if m and int(m.group(2)) == waves:
            r[m.group(1)] = float(m.group(3))
    return r


def mnemonic_cycles(rates):
    exact, wild = {}, []
    for k, ms in RATE_KERNELS.items():
        if k not in rates:
            continue
        for m in ms:
            if "*" in m:
                wild.append((re.compile("^" + m.replace("*", r"\w+") + "$"), rates[k], k))
            else:
                exact[m] = (rates[k], k)
    return exact, wild


def base_mnemonic(tok):
    return re.sub(r"_(e32|e64)$", "", tok)


def price(tok, exact, wild, rates):
    """(cycles, source) of one vector instruction token"""
    for suf, k in SUFFIX_KERNELS.items():
        if tok.endswith(suf) and k in rates:
            return rates[k], k
    b = base_mnemonic(re.sub(r"_(sdwa|dpp)$", "", tok))
    if b in exact:
        return exact[b]
    for rx, c, k in wild:
        if rx.match(b):
            return c, k
    return DEFAULT_CYCLES, "default"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] for o in out]


def function_lines(asm_path, kernel):
    lines = open(asm_path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    names = demangle([s for _, s in starts])
    for (i, sym), name in zip(starts, names):
        if name.replace(" ", "") == kernel.replace(" ", ""):
            j = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
            return lines[i + 1:j], sym
    sys.exit(f"kernel {kernel!r} not found in {asm_path}; has: " + ", ".join(sorted(set(names)))[:2000])


def blocks_of(lines):
    """[(label, [instruction tokens with operands])]; a new block at every label and at every `; %bb.N:` marker"""
    blocks, cur = [], ("entry", [])
    for l in lines:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        m2 = re.match(r"^; %bb\.(\d+):", s)
        if m or m2:
            blocks.append(cur)
            cur = (m.group(1) if m else f"bb.{m2.group(1)}", [])
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur[1].append(s.split(";")[0].strip())
    blocks.append(cur)
    return blocks


def find_loop(blocks, needle):
    """innermost backward-branch loop that contains an instruction starting with `needle`: (first block index, last block index)"""
    pos = {lab: i for i, (lab, _) in enumerate(blocks)}
    loops = []
    for i, (lab, ins) in enumerate(blocks):
        for x in ins:
            m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", x)
            if m and m.group(1) in pos and pos[m.group(1)] <= i:
                loops.append((pos[m.group(1)], i))
    cands = [(a, b) for a, b in loops if any(x.startswith(needle) for _, ins in blocks[a:b + 1] for x in ins)]
    if not cands:
        sys.exit(f"no loop with {needle}")
    a, b = min(cands, key=lambda ab: ab[1] - ab[0])
    # the latch blocks of a loop may be laid out behind one another, each with its own backward branch: extend to the last block that
    # branches back into [a, b] without leaving through an enclosing loop's header
    grown = True
    while grown:
        grown = False
        for a2, b2 in loops:
            if a <= a2 <= b and b2 > b:      # (an enclosing loop's header lies before a: its backward branch is not taken for one of ours)
                b, grown = b2, True
    return a, b


def classify(tok):
    if tok.startswith("v_"):
        return "valu"
    if tok.startswith("s_"):
        if tok.startswith(("s_waitcnt", "s_nop")):
            return "wait"
        if tok.startswith(("s_cbranch", "s_branch")):
            return "branch"
        if tok.startswith(("s_load", "s_buffer_load")):
            return "smem"
        return "salu"
    if tok.startswith("ds_"):
        return "lds"
    if tok.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"



REPLAY_TEMPLATE = r"""// GENERATED by tools/isa_hist.py --emit-replay from %(asm)s: the vector and scalar-ALU instructions of the hot loop of
// %(kernel)s (blocks %(blocks)s), in the compiler's own order and registers, WITHOUT its memory / LDS instructions, waits and branches.
// Every wavefront runs the sequence ITER times between two s_memtime reads; the grid fills every SIMD of the chip to the kernel's own occupancy.
// SIMD cycles per iteration and wavefront = wavefront cycles / wavefronts per SIMD: the time the vector pipe needs for THIS instruction
// sequence when nothing else stalls -- the issue roof bench.py prices the kernel against (roofline.issue_*).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define ITER %(iter)d
__global__ __launch_bounds__(%(threads)d) __attribute__((amdgpu_waves_per_eu(%(wpe)d, %(wpe)d))) void k_replay(unsigned long long* out) {
    unsigned lo, hi;
    asm volatile(
        "s_memtime s[%(t0)d:%(t0b)d]\n s_waitcnt lgkmcnt(0)\n s_movk_i32 s%(cnt)d, ITER_\n"
%(body)s
        "99:\n"
        "s_memtime s[%(t1)d:%(t1b)d]\n s_waitcnt lgkmcnt(0)\n"
        "s_sub_u32 %%0, s%(t1)d, s%(t0)d\n s_subb_u32 %%1, s%(t1b)d, s%(t0b)d\n"
        : "=s"(lo), "=s"(hi) : "s"(__builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) & 7u))) : %(clobbers)s);
    if ((threadIdx.x & 63) == 0) out[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = ((unsigned long long)hi << 32) | lo;
}
int main(int argc, char** argv) {
    const double sclk_hz = argc > 1 ? atof(argv[1]) * 1e6 : 2.4e9;     // shader clock the event-timed figure is converted with (MHz on the command line)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_replay, %(threads)d, 0) != hipSuccess || per_cu < 1) return 2;
    const int cap = per_cu * prop.multiProcessorCount, wpb = %(threads)d / 64, rounds = 8;
    unsigned long long* d;
    if (hipMalloc(&d, sizeof(unsigned long long) * cap * rounds * wpb) != hipSuccess) return 3;
    // (a) one resident set: every wavefront's own s_memtime span; the SLOWEST wavefront of a SIMD that arbitrates oldest-first has waited for all the
    //     others, so its span / wavefronts per SIMD is the SIMD's time per wavefront-iteration
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_replay, dim3(cap), dim3(%(threads)d), 0, 0, d); if (hipDeviceSynchronize() != hipSuccess) return 4; }
    std::vector<unsigned long long> h(cap * wpb);
    if (hipMemcpy(h.data(), d, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost) != hipSuccess) return 5;
    std::sort(h.begin(), h.end());
    const double wps = per_cu * wpb / 4.0, p99 = (double)h[(size_t)(h.size() * 0.99)], mx = (double)h.back(), mn = (double)h.front();
    // (b) eight resident sets back to back, timed by HIP events: SIMD cycles = time x clock x SIMDs / wavefront-iterations (no assumption on arbitration)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_replay, dim3(cap * rounds), dim3(%(threads)d), 0, 0, d);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_replay, dim3(cap * rounds), dim3(%(threads)d), 0, 0, d);
    hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) return 6;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double simds = prop.multiProcessorCount * 4.0;
    const double cyc_evt = ms * 1e-3 * sclk_hz * simds / ((double)cap * rounds * wpb * ITER);
    printf("{\"kernel\": \"%(kernel_js)s\", \"valu_per_iteration\": %(nvalu)d, \"salu_per_iteration\": %(nsalu)d, \"waves_per_simd\": %%.2f, \"iterations\": %%d, "
           "\"simd_cycles_per_iteration_events\": %%.2f, \"events_ms\": %%.4f, \"sclk_mhz_assumed\": %%.0f, "
           "\"simd_cycles_per_iteration_slowest_wave\": %%.2f, \"simd_cycles_per_iteration_p99_wave\": %%.2f, \"simd_cycles_per_iteration_fastest_wave\": %%.2f, "
           "\"cycles_per_valu_inst_events\": %%.4f}\n", wps, ITER, cyc_evt, ms, sclk_hz / 1e6, mx / ITER / wps, p99 / ITER / wps, mn / ITER / wps, cyc_evt / %(nvalu)d.0);
    return 0;
}
"""


def emit_replay(o, loop_blocks, weights, rows):
    """the loop's instruction sequence as one inline-asm block (see REPLAY_TEMPLATE)"""
    order = o.replay_order.split(",") if o.replay_order else [lab for lab, _ in loop_blocks]
    by = dict(loop_blocks)
    body, vregs, sregs, used_blocks = [], set(), set(), []
    n_valu = n_salu = 0
    for lab in order:
        if weights.get(lab, 1.0) < o.replay_min_weight:
            continue
        used_blocks.append(lab)
        for x in by[lab]:
            if x.startswith(";;#") or x.startswith("#"):
                continue
            tok = x.split()[0]
            kind = classify(tok)
            if o.replay_pad and (kind in ("lds", "vmem", "wait", "branch") or (kind == "salu" and (re.search(r"\bexec\b", x.split(None, 1)[1].split(",")[0]) or tok.startswith(("s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec"))))):
                body.append('        "s_nop 0\\n"')          # the instruction's place in the wavefront's stream, without its effect
                continue
            if kind not in ("valu", "salu") or (kind == "salu" and o.replay_no_salu):
                continue
            if kind == "salu" and (re.search(r"\bexec\b", x.split(None, 1)[1].split(",")[0]) or tok.startswith(("s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec", "s_setpc", "s_getpc", "s_endpgm", "s_barrier", "s_sleep", "s_setprio", "s_memtime", "s_memrealtime"))):
                continue                                   # (nothing may change exec: every lane stays on)
            if tok.startswith(("v_readlane", "v_writelane", "v_readfirstlane")) and "exec" in x:
                continue
            x2 = x.replace("%", "%%")
            body.append(f'        "{x2}\\n"')
            n_valu += kind == "valu"
            n_salu += kind == "salu"
            for m in re.finditer(r"\bv(\d+)\b", x):
                vregs.add(int(m.group(1)))
            for m in re.finditer(r"\bv\[(\d+):(\d+)\]", x):
                vregs.update(range(int(m.group(1)), int(m.group(2)) + 1))
            for m in re.finditer(r"\bs(\d+)\b", x):
                sregs.add(int(m.group(1)))
            for m in re.finditer(r"\bs\[(\d+):(\d+)\]", x):
                sregs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    smax = max(sregs | {11})
    t0 = (smax + 2) & ~1
    cnt, t1 = t0 + 2, t0 + 4
    if t1 + 1 > 101:
        sys.exit("replay: no scalar registers left for the timer")
    sregs |= {t0, t0 + 1, cnt, t1, t1 + 1}
    clob = ['"vcc"', '"scc"', '"memory"'] + [f'"v{i}"' for i in sorted(vregs)] + [f'"s{i}"' for i in sorted(sregs)]
    nv = max(vregs) + 1
    wpe = max(1, min(8, 512 // (((nv + 4) + 7) // 8 * 8)))       # the replay's own occupancy: the body's registers + the few the compiler adds
    threads = 512
    # --replay-rotate: the eight wavefronts of a workgroup (two per SIMD, four workgroups per CU) each start the sequence at another eighth of it, as
    # the wavefronts of the real kernel are at different places of the loop at any time; without it every wavefront of a SIMD asks for the same
    # kind of instruction at the same time
    nrot = 8 if o.replay_rotate else 1
    variants = []
    for r in range(nrot):
        k = (len(body) * r) // nrot
        rot = body[k:] + body[:k]
        head = [f'        "s_cmp_lg_u32 %2, {r}\\n s_cbranch_scc1 {r + 11}f\\n"'] if nrot > 1 else []
        variants.append("\n".join(head + [f'        "{r + 1}:\\n"'] + rot +
                                  [f'        "s_sub_u32 s{cnt}, s{cnt}, 1\\n s_cmp_lg_u32 s{cnt}, 0\\n s_cbranch_scc1 {r + 1}b\\n s_branch 99f\\n"'] +
                                  ([f'        "{r + 11}:\\n"'] if nrot > 1 else [])))
    src = REPLAY_TEMPLATE % {"asm": o.asm, "kernel": o.kernel, "kernel_js": o.kernel.replace('"', "'"), "blocks": " ".join(used_blocks), "iter": 3000, "threads": threads,
                             "wpe": wpe, "t0": t0, "t0b": t0 + 1, "cnt": cnt, "t1": t1, "t1b": t1 + 1, "body": "\n".join(variants), "clobbers": ", ".join(clob),
                             "nvalu": n_valu, "nsalu": n_salu}
    src = src.replace("ITER_", "3000")
    with open(o.emit_replay, "w") as f:
        f.write(src)
    return {"file": o.emit_replay, "blocks": used_blocks, "valu": n_valu, "salu": n_salu, "vgprs_of_body": nv}


def auto_weights(loop_blocks, freq, bins):
    """k_sample<2>'s float32 pool loop: which of its blocks run how often per iteration, from what the blocks contain and the frequencies the stamped
    build counted (tools/stamps.py 2).  Returns ({label: weight}, {label: why})."""
    w, why = {}, {}
    period = 0.25                                                    # VGL_SLOW_PERIOD 4: the ballot that asks whether any lane needs a bounded test
    region = None                                                    # inside a bounded test's blocks until the join block (starts with s_or_b64 exec)
    for lab, ins in loop_blocks:
        toks = [x.split()[0] for x in ins if not x.startswith(";;#")]
        valu = [t for t in toks if t.startswith("v_")]
        text = " ".join(ins)
        if region and toks and toks[0] == "s_or_b64" and "exec" in ins[0]:
            region = None
        if region:
            w[lab], why[lab] = freq[region], f"inside the bounded {region.split('_')[0]} test ({freq[region]:.3f} per iteration, measured)"
            continue
        if "global_atomic" in text or (valu and all(t.startswith("v_readlane") or t.startswith("v_mov") for t in valu) and "v_readlane_b32" in toks):
            w[lab], why[lab] = 0.0, "error flag (never at the bench configurations)"
        elif len(valu) == 2 and toks[-1].startswith("s_cbranch_vcc") and valu[0].startswith("v_cndmask") and valu[1].startswith("v_cmp_ne_u32"):
            w[lab], why[lab] = period, "does any lane need a bounded test? (every fourth iteration)"
        elif sum(t.startswith("v_fma_f32") for t in valu) >= 8:
            region = "gamma_test"
            w[lab], why[lab] = freq[region], f"bounded gamma test ({freq[region]:.3f} per iteration, measured)"
        elif "v_log_f32_e32" in toks and any(t.startswith("v_rcp_f32") for t in toks) and "v_cvt_i32_f32_e32" not in toks and len(valu) <= 16:
            region = "normal_test"
            w[lab], why[lab] = freq[region], f"bounded normal-deviate test ({freq[region]:.3f} per iteration, measured)"
        elif "ds_read_u8" in toks:
            w[lab], why[lab] = (freq["finish"] if bins else 0.0), "finish path, --qs-bins table lookup"
        elif len(valu) <= 2 and any(t.startswith("v_min_i32_sdwa") for t in toks) and toks[-1] == "s_branch":
            w[lab], why[lab] = (0.0 if bins else freq["finish"]), "finish path without --qs-bins (cap at 63)"
        elif len(valu) <= 2 and valu and all(t.startswith(("v_cndmask", "v_and_b32")) for t in valu) and toks[-1] == "s_branch":
            w[lab], why[lab] = (freq["finish"] if bins else 0.0), "finish path, --qs-bins (no bin -> 0)"
        elif "v_cvt_i32_f32_e32" in toks or "ds_add_rtn_u32" in toks or "ds_write_b16" in toks:
            w[lab], why[lab] = freq["finish"], f"a lane finishes a read ({freq['finish']:.3f} per iteration, measured)"
    return w, why

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm", required=True)
    ap.add_argument("--kernel", required=True, help="demangled name without arguments, e.g. 'k_sample<2, false, 1, false, 2>'")
    ap.add_argument("--loop-with", default="ds_add_rtn_u32", help="the loop = the innermost one containing an instruction that starts with this")
    ap.add_argument("--rates", required=True)
    ap.add_argument("--weight", action="append", default=[], help="LABEL=w: how often the block runs per iteration (default 1)")
    ap.add_argument("--note", action="append", default=[], help="LABEL=why (printed with the weight)")
    ap.add_argument("--auto", default=None, help="JSON written by tools/stamps.py 2 (per_iteration: normal_test, gamma_test, finish): classify the pool loop's blocks of "
                                                 "k_sample<2> by what they contain and weigh them with the measured frequencies (printed; --weight still overrides)")
    ap.add_argument("--bins", type=int, default=0, help="--auto: 1 = the workload has --qs-bins (the table-lookup block of the finish path runs), 0 = it has not")
    ap.add_argument("--list", action="store_true", help="print the loop's blocks with their first instructions (to choose the weights)")
    ap.add_argument("--json", default=None)
    ap.add_argument("--emit-replay", default=None, help="write a HIP program that replays the loop's vector (and scalar ALU) instructions of the blocks with "
                                                        "weight >= --replay-min-weight, without memory, LDS, waits or branches: its time per iteration at full occupancy IS "
                                                        "the issue roof of this instruction sequence (tools/isa_replay.md)")
    ap.add_argument("--replay-min-weight", type=float, default=0.5)
    ap.add_argument("--replay-no-salu", action="store_true", help="vector instructions only")
    ap.add_argument("--replay-pad", action="store_true", help="an `s_nop 0` in the place of every memory / LDS / wait / branch / exec-writing instruction")
    ap.add_argument("--replay-rotate", action="store_true", help="each of a workgroup's eight wavefronts starts at another eighth of the sequence")
    ap.add_argument("--replay-order", default=None, help="comma-separated block labels: the order the replay runs the blocks in (default: layout order)")
    o = ap.parse_args()

    rates = load_rates(o.rates)
    exact, wild = mnemonic_cycles(rates)
    lines, sym = function_lines(o.asm, o.kernel)
    blocks = blocks_of(lines)
    a, b = find_loop(blocks, o.loop_with)
    notes = dict(n.split("=", 1) for n in o.note)
    weights = {}
    if o.auto:
        weights, auto_notes = auto_weights(blocks[a:b + 1], json.load(open(o.auto))["per_iteration"], o.bins)
        for k, v in auto_notes.items():
            notes.setdefault(k, v)
    weights.update({k: float(v) for k, v in (w.split("=") for w in o.weight)})
    unknown_labels = [k for k in weights if k not in [lab for lab, _ in blocks[a:b + 1]]]
    if unknown_labels:
        sys.exit(f"--weight names blocks that are not in the loop: {unknown_labels}")

    per_class, per_kind = {}, {}
    tot_c = tot_n = 0.0
    unpriced = {}
    rows = []
    for lab, ins in blocks[a:b + 1]:
        w = weights.get(lab, 1.0)
        n_v = 0
        c_v = 0.0
        for x in ins:
            if x.startswith(";;#") or x.startswith("#"):
                continue
            tok = x.split()[0]
            kind = classify(tok)
            per_kind[kind] = per_kind.get(kind, 0.0) + w
            if kind != "valu":
                continue
            c, src = price(tok, exact, wild, rates)
            if src == "default":
                unpriced[tok] = unpriced.get(tok, 0) + 1
            key = base_mnemonic(tok)
            e = per_class.setdefault(key, {"count": 0.0, "cycles_each": c, "priced_by": src})
            e["count"] += w
            n_v += 1
            c_v += c
        tot_c += w * c_v
        tot_n += w * n_v
        rows.append({"block": lab, "weight": w, "valu": n_v, "cycles": round(c_v, 2), "why": notes.get(lab, "every iteration" if w == 1.0 else "")})
        if o.list:
            print(f"{lab:12s} w={w:<5g} valu={n_v:3d} cyc={c_v:7.1f}  | " + " ; ".join(i.split()[0] for i in ins[:7]))

    res = {"kernel": o.kernel, "symbol": sym, "loop": {"first_block": blocks[a][0], "last_block": blocks[b][0], "contains": o.loop_with},
           "rates_file": o.rates, "rates_waves_per_simd": 4,
           "valu_insts_per_iteration": round(tot_n, 2), "issue_cycles_per_iteration": round(tot_c, 1),
           "issue_cycles_per_inst": round(tot_c / tot_n, 4), "other_per_iteration": {k: round(v, 2) for k, v in sorted(per_kind.items())},
           "blocks": rows,
           "classes": {k: {"count": round(v["count"], 2), "cycles_each": v["cycles_each"], "cycles": round(v["count"] * v["cycles_each"], 1), "priced_by": v["priced_by"]}
                       for k, v in sorted(per_class.items(), key=lambda kv: -kv[1]["count"] * kv[1]["cycles_each"])},
           "unpriced_mnemonics_at_default": unpriced, "default_cycles": DEFAULT_CYCLES}
    if o.emit_replay:
        res["replay"] = emit_replay(o, blocks[a:b + 1], weights, rows)
    if o.json:
        with open(o.json, "w") as f:
            json.dump(res, f, indent=1)
    if not o.list:
        print(f"{o.kernel}: loop {blocks[a][0]} .. {blocks[b][0]}: {tot_n:.1f} vector instructions, {tot_c:.0f} issue cycles per iteration "
              f"= {tot_c / tot_n:.3f} cycles per instruction; other: {res['other_per_iteration']}")
        for k, v in list(res["classes"].items())[:40]:
            print(f"  {k:24s} x{v['count']:6.2f}  {v['cycles_each']:5.2f} cyc  = {v['cycles']:7.1f}   ({v['priced_by']})")
        if unpriced:
            print("  priced at the default", DEFAULT_CYCLES, ":", unpriced)


if __name__ == "__main__":
    main()
