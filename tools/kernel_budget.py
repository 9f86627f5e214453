#!/usr/bin/env python3
"""tools/kernel_budget.py -- register / spill / scratch / occupancy / LDS of every kernel instantiation the shipped library is built from,
as numbers, against the committed budget profiles/kernel_budget.json.

Why: round 5 lost 4 % on the depth-5 configuration to a two-line edit of a test hook in code that configuration never runs -- the
edit cost the fused kernel 51 scalar-register spills.  Nothing in the suite saw it.  tests/test_kernel_budget_cpu.py now compiles the
device code of every kernel file exactly as the Makefile does (hipcc cross-compiles gfx950 without a GPU), reads the compiler's
kernel-resource-usage remarks and fails when an instantiation needs MORE registers, spills, scratch or LDS, or reaches FEWER wavefronts
per SIMD, than the committed budget says -- or when a kernel appears or disappears without the budget file being updated.

usage: python tools/kernel_budget.py            -> compare, exit 1 on a regression
       python tools/kernel_budget.py --update   -> rewrite profiles/kernel_budget.json from the working tree
       python tools/kernel_budget.py --print    -> the current numbers as a table
"""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vcfgl_amd", "csrc")
BUDGET = os.path.join(ROOT, "profiles", "kernel_budget.json")
FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
          "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}
# a build may need at most the budget of these, and must reach at least the budget's occupancy
LOWER_IS_BETTER = ("vgprs", "agprs", "scratch", "sgpr_spill", "vgpr_spill", "lds")


def makefile_flags():
    """CXXFLAGS and the kernel files, read from the Makefile so that the test compiles what the library is built from"""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    arch = re.search(r"^ARCH\s*\?=\s*(\S+)", mk, re.M).group(1)
    flags = re.search(r"^CXXFLAGS\s*=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", arch).split()
    kernels = re.search(r"^KERNELS\s*=\s*(.*)$", mk, re.M).group(1).split()
    hipcc = os.environ.get("HIPCC") or re.search(r"^HIPCC\s*\?=\s*(\S+)", mk, re.M).group(1)
    return hipcc, flags, kernels


def extra_flags(kernel):
    """per-file options of the Makefile (SEGFLAGS: vgl_sample_seg.hip is compiled with scheduler options of its own)"""
    if kernel != "vgl_sample_seg":
        return []
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^SEGFLAGS\s*\?=[ \t]*(.*)$", mk, re.M)
    return m.group(1).split() if m else []


def remarks_of(hipcc, flags, kernel, outdir):
    os.makedirs(outdir, exist_ok=True)
    cmd = [hipcc] + flags + extra_flags(kernel) + ["-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(outdir, kernel + ".o"), kernel + ".hip"]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed:\n{r.stderr[-2000:]}")
    return r.stderr


def parse(remarks):
    rows, cur = {}, None
    for l in remarks.split("\n"):
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = rows.setdefault(m.group(1), {})
            continue
        # ("remark: file:line:col:     VGPRs: 64" with -save-temps, "file:line:col: remark:     VGPRs: 64" without)
        m = re.search(r"remark:.*?\s(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", l)
        if m and cur is not None:
            cur[FIELDS[m.group(1)]] = int(m.group(2))
    return rows


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] for o in out]


def current(outdir=None):
    hipcc, flags, kernels = makefile_flags()
    outdir = outdir or os.path.join(ROOT, "build", "budget")
    with ThreadPoolExecutor(max_workers=min(4, os.cpu_count() or 1)) as ex:
        texts = list(ex.map(lambda k: remarks_of(hipcc, flags, k, outdir), kernels))
    res = {}
    for k, t in zip(kernels, texts):
        rows = parse(t)
        for sym, name in zip(rows, demangle(list(rows))):
            res[name] = dict(rows[sym], file=k + ".hip")
    return res


def compare(cur, budget):
    """list of human-readable violations"""
    bad = []
    for name in sorted(set(cur) | set(budget)):
        if name not in budget:
            bad.append(f"{name}: a new kernel instantiation without a budget (python tools/kernel_budget.py --update, and look at its numbers)")
            continue
        if name not in cur:
            bad.append(f"{name}: in the budget file, no longer built (python tools/kernel_budget.py --update)")
            continue
        c, b = cur[name], budget[name]
        for f in LOWER_IS_BETTER:
            if c.get(f, 0) > b.get(f, 0):
                bad.append(f"{name}: {f} {b.get(f, 0)} -> {c.get(f, 0)}")
        if c.get("occupancy", 0) < b.get("occupancy", 0):
            bad.append(f"{name}: occupancy {b['occupancy']} -> {c['occupancy']} wavefronts per SIMD")
    return bad


def main():
    cur = current()
    if "--print" in sys.argv:
        for n, v in sorted(cur.items()):
            print(f"{n:48s} " + " ".join(f"{k}={v[k]}" for k in ("vgprs", "sgprs", "sgpr_spill", "vgpr_spill", "scratch", "occupancy", "lds") if k in v))
        return
    if "--update" in sys.argv:
        with open(BUDGET, "w") as f:
            json.dump({"note": "compiler's kernel-resource-usage remarks of every kernel instantiation of the shipped library (tools/kernel_budget.py --update); "
                               "tests/test_kernel_budget_cpu.py fails on more registers / spills / scratch / LDS or fewer wavefronts per SIMD than listed",
                       "kernels": cur}, f, indent=1, sort_keys=True)
        print(f"{BUDGET}: {len(cur)} kernels")
        return
    bad = compare(cur, json.load(open(BUDGET))["kernels"])
    for b in bad:
        print(b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
