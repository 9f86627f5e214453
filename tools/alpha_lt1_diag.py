#!/usr/bin/env python3
"""Diagnostic: --error-qs 2 with a beta shape alpha < 1 (rng.h:146-148, the pow() branch of the gamma sampler):
per-read base / quality score and the beta deviate itself, HIP path against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib, synth
from vcfgl_amd import Simulator, VcfglArgs, _abi

S, N, CAP = int(os.environ.get("S", "400")), 64, 24
for (mean, var) in ((0.4, 0.1), (0.3, 0.15), (0.05, 0.03)):
    a = VcfglArgs(seed=3, depth=4, error_rate=mean, error_qs=2, beta_variance=var)
    a.rng_mode, a.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    gt = synth.binary_sites(0, S, N)
    want = oracle_lib.Oracle(a, N).simulate(0, gt, read_capacity=CAP, deviates=True)
    sim = Simulator(a, N, max_sites_per_tile=S)
    got = sim.simulate(0, gt, read_capacity=CAP, deviates=True)
    sim.close()
    wr, gr = want.numpy("reads"), got.numpy("reads")
    we, ge = want.numpy("read_errp"), got.numpy("read_errp")
    have = wr != 0xFF
    bad = np.argwhere((wr != gr) & have)
    eb = np.argwhere((we.view(np.uint64) != ge.view(np.uint64)) & have)
    print(f"beta({mean},{var}): reads {int(have.sum())}, read bytes differ {len(bad)}, deviates differ (bitwise) {len(eb)}")
    for r, s, k in eb[:12]:
        x, y = we[r, s, k], ge[r, s, k]
        print(f"   read {r} site {s} sample {k}: oracle p {x!r} device p {y!r} rel {abs(x-y)/max(abs(x),1e-300):.3e}  q oracle {wr[r,s,k]>>2} device {gr[r,s,k]>>2}")
    for r, s, k in bad[:12]:
        print(f"   BYTE read {r} site {s} sample {k}: oracle {wr[r,s,k]:#x} device {gr[r,s,k]:#x} p oracle {we[r,s,k]!r} device {ge[r,s,k]!r}")
