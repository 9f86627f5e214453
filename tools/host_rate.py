#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point vgl_simulate_tile (pageable numpy buffers,
synchronous copies) on the C3 flags -- the number DESIGN.md quotes beside the device-resident bench."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, synth
from vcfgl_amd import Simulator, VcfglArgs
N, S = 1000, 8192
a = VcfglArgs(seed=42, depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5)
sim = Simulator(a, N, max_sites_per_tile=S)
gt = synth.binary_sites(0, S, N)
tile = sim.new_tile(S, fields=["fmt_dp", "gl"])
import ctypes as C
sim._check(sim.lib.vgl_simulate_tile(sim.ctx, 0, S, gt.ctypes.data, tile.byref()))
t = time.time(); n = 4
for i in range(n):
    sim._check(sim.lib.vgl_simulate_tile(sim.ctx, (i + 1) * S, S, gt.ctypes.data, tile.byref()))
dt = (time.time() - t) / n
print(f"host-buffer tile of {S}x{N}: {dt*1e3:.1f} ms -> {S*N/dt:.3e} evals/s, {S*N*65/dt/1e9:.1f} GB/s over PCIe")
