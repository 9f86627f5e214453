"""Where the path sits against the HBM roof as a function of depth: whole-tile kernel time of one 32768-site x 1000-sample tile
(default tags GL + DP, G = 15: 65 algorithmic bytes per evaluation), fixed quality score and --error-qs 2, depth 1 ... 60.
usage (GPU box): python tools/depth_sweep.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from vcfgl_amd import Simulator, VcfglArgs
N, S, B = 1000, 32768, 65.0
gt = synth.binary_sites_torch(0, S, N, "cuda:0")
print("error-qs depth  ms/tile (depth, sample, redo, site, gl, siteagg)            evals/s   algorithmic GB/s  of 8 TB/s")
for eqs in (0, 2):
    for depth in (1.0, 2.0, 5.0, 10.0, 20.0, 30.0, 60.0):
        kw = dict(error_qs=2, beta_variance=1e-5) if eqs == 2 else {}
        sim = Simulator(VcfglArgs(seed=42, depth=depth, error_rate=0.01, **kw), N, max_sites_per_tile=S)
        sim.timing(True)
        tile = sim.new_tile(S, fields=["fmt_dp", "gl"], device="cuda:0")
        sim.simulate_device(0, gt, tile); sim.check(); sim.kernel_ms(reset=True)
        for _ in range(3):
            sim.simulate_device(0, gt, tile); sim.check()
        ms, n = sim.kernel_ms(reset=True)
        per = [x / max(k, 1) for x, k in zip(ms, n)]
        t = sum(per) * 1e-3
        print(f"{eqs:8d} {depth:5.0f}  {[round(x, 3) for x in per]!s:42s} {S * N / t:10.3e} {S * N * B / t / 1e9:12.0f} {S * N * B / t / 8e12:10.3f}", flush=True)
        sim.close(); del tile
