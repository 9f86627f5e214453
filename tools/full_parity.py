#!/usr/bin/env python3
"""One-off: a bench configuration at its FULL size against the CPU oracle, site by site and field by field (the suite's
tests/test_gpu_scale_oracle.py does 1e8 of C3's 1e9 evaluations).  The oracle is test infrastructure; this script is a checker.
usage (GPU box): python tools/full_parity.py [c3|c4|c5|alltags|qsi16|fq20|c5wide|precise|fixedq|gl1q] [sites]      -- prints a progress line per 100 000 sites"""
import dataclasses
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_pool
import test_gpu_scale_oracle as T

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
full = {"c3": 1_000_000, "c4": 1_250_000, "c5": 2_000_000, "alltags": 262_144, "qsi16": 262_144, "fq20": 1_000_000, "c5wide": 1_000_000, "precise": 262_144, "fixedq": 1_000_000, "gl1q": 262_144}[name]
case = dict(T.CASES[name], S=int(sys.argv[2]) if len(sys.argv) > 2 else full, site0=0)
t0 = time.time()
got, sim = T._gpu_checksums(case)
sim.close()
print(f"{name}: device checksums of {case['S']} sites x {case['N']} samples in {time.time() - t0:.1f} s", flush=True)
args = dataclasses.asdict(T._args(case["flags"]))
bad = 0
step = 100_000 if case["N"] <= 1000 else 25_000
for s0 in range(0, case["S"], step):
    n = min(step, case["S"] - s0)
    want = oracle_pool.oracle_site_checksums(args, case["N"], s0, n, case["fields"], gt=case["gt"])
    d = np.argwhere(got[s0:s0 + n] != want)
    bad += len(d)
    print(f"  sites [{s0}, {s0 + n}): {len(d)} differing (site, field) checksums, {time.time() - t0:.0f} s", flush=True)
    if len(d):
        print("   first:", s0 + int(d[0][0]), case["fields"][int(d[0][1])])
print(f"{name}: {case['S'] * case['N']:.3g} evaluations, {len(case['fields'])} fields: {'ALL EQUAL to the oracle' if bad == 0 else str(bad) + ' DIFFER'}")
sys.exit(1 if bad else 0)
