#!/usr/bin/env python3
"""Diagnostic: how many reads of a tile k_sample<2, deferred> hands to k_redo (the float32 bounds could not settle them).
usage (GPU box): python tools/redo_rate.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from vcfgl_amd import Simulator, VcfglArgs
for name, kw, N, S in (("c3", dict(depth=20.0, beta_variance=1e-5), 1000, 16384), ("c4", dict(depth=30.0, beta_variance=1e-5), 2000, 8192),
                       ("var 1e-4", dict(depth=20.0, beta_variance=1e-4), 1000, 16384), ("e 0.05 var 2e-4", dict(depth=20.0, error_rate=0.05, beta_variance=2e-4), 1000, 16384)):
    a = VcfglArgs(seed=42, error_qs=2, **({"error_rate": 0.01} | kw))
    sim = Simulator(a, N, max_sites_per_tile=S, hooks=True)
    gt = synth.binary_sites_torch(0, S, N, "cuda:0")
    tile = sim.new_tile(S, fields=["fmt_dp", "gl"], device="cuda:0")
    sim.simulate_device(0, gt, tile); sim.check()
    n = C.c_uint(0)
    sim.lib.vgl_dbg_redo_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    assert sim.lib.vgl_dbg_redo_count(sim.ctx, C.byref(n)) == 0
    reads = float(tile.numpy("fmt_dp").sum()) if hasattr(tile, "numpy") else S * N * kw["depth"]
    print(f"{name}: {n.value} of {reads:.3g} reads redone = {n.value / reads:.2e}")
    sim.close()
