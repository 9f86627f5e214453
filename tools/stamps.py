#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_sample<2> (in-kernel clock64 stamps, VGL_DEBUG_STAMPS=1).
Shares only -- a stamped run is never a timing result."""
import ctypes as C, os, sys
os.environ["VGL_DEBUG_STAMPS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from vcfgl_amd import Simulator, VcfglArgs, _abi
N, S = 1000, 8192
a = VcfglArgs(seed=42, depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5)
sim = Simulator(a, N, max_sites_per_tile=S, hooks=True)
gt = synth.binary_sites_torch(0, S, N, "cuda:0")
tile = sim.new_tile(S, fields=["fmt_dp", "gl"], device="cuda:0")
sim.simulate_device(0, gt, tile); sim.check()
out = (C.c_ulonglong * 16)()
sim.lib.vgl_dbg_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert sim.lib.vgl_dbg_stamps(sim.ctx, out) == 0
w = out[0]
names = ["waves", "cycles/wave", "streams+poisson", "owner(bases)", "pool(beta)", "flush", "pool iterations", "pool items"]
for n, v in zip(names, out):
    print(f"{n:18s} {v / w:12.1f}" if n != "waves" else f"{n:18s} {v}")
