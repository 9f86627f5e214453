#!/usr/bin/env python3
"""Throughput of VGL_RNG_SERIAL (reference draw order) on the device, device-resident tiles."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from vcfgl_amd import Simulator, VcfglArgs, _abi
cases = [("depth 10 GL1 fixed-q (C2 flags)", dict(depth=10, error_rate=0.01, gl_model=1), 4000, 100),
         ("depth 20 GL2 fixed-q, 1000 samples", dict(depth=20, error_rate=0.01), 400, 1000),
         ("depth 20 error-qs 2 std::mt19937 beta", dict(depth=20, error_rate=0.01, error_qs=2, beta_variance=1e-5), 100, 100),
         ("depth 20 error-qs 2 std beta, 1000 samples", dict(depth=20, error_rate=0.01, error_qs=2, beta_variance=1e-5), 1000, 1000)]
for name, kw, S, N in cases:
    a = VcfglArgs(seed=42, **kw); a.rng_mode = _abi.VGL_RNG_SERIAL; a.beta_sampler = _abi.VGL_BETA_STD
    sim = Simulator(a, N, max_sites_per_tile=S)
    gt = synth.binary_sites_torch(0, 2 * S, N, "cuda:0")
    tile = sim.new_tile(S, fields=["fmt_dp", "gl"], device="cuda:0")
    sim.simulate_device(0, gt[:S], tile); sim.check()
    torch.cuda.synchronize(); t = time.time()
    sim.simulate_device(S, gt[S:], tile); sim.check()
    dt = time.time() - t
    print(f"{name:42s} {S * N / dt:12.0f} evals/s")
    sim.close()
