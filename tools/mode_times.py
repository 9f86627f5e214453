"""Diagnostic: per-kernel time of one 16384-site x 1000-sample tile at the C3 flags for the default surface and three optional ones.
usage (GPU box): python tools/mode_times.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from vcfgl_amd import Simulator, VcfglArgs
N, S = 1000, 16384
for name, kw in (("plain", {}), ("precise-gl 1", dict(precise_gl=1)), ("addGP addPL", dict(add_gp=1, add_pl=1)), ("GL1 per-read q", dict(gl_model=1))):
    a = VcfglArgs(seed=42, depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, **kw)
    sim = Simulator(a, N, max_sites_per_tile=S)
    sim.timing(True)
    gt = synth.binary_sites_torch(0, S, N, "cuda:0")
    fields = ["fmt_dp", "gl"] + (["gp", "pl"] if "add_gp" in kw else [])
    tile = sim.new_tile(S, fields=fields, device="cuda:0")
    for _ in range(3):
        sim.simulate_device(0, gt, tile); sim.check()
    ms, n = sim.kernel_ms(reset=True)
    print(name, [round(x / max(k, 1), 3) for x, k in zip(ms, n)], "ms per 16384-site tile (depth, sample, redo, site, gl, siteagg)")
    sim.close()
