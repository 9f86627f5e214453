"""k_gl against k_gl2 (hooks build, VGL_GL2X=0 / 1) as a function of depth and sample count: k_gl's time per 32768-site tile, default tags.
usage (GPU box): python tools/gl2x_sweep.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from vcfgl_amd import Simulator, VcfglArgs
print("error-qs     N depth   k_gl ms (VGL_GL2X=0)   k_gl2 ms (VGL_GL2X=1)   ratio")
for eqs, N, S, depths in ((0, 1000, 32768, (12.0, 16.0, 20.0, 30.0, 60.0)), (2, 1000, 32768, (16.0, 20.0, 24.0, 28.0, 30.0, 40.0)), (2, 2000, 16384, (20.0, 30.0)), (0, 500, 32768, (20.0,))):
    gt = synth.binary_sites_torch(0, S, N, "cuda:0")
    for depth in depths:
        kw = dict(error_qs=2, beta_variance=1e-5) if eqs == 2 else {}
        t = []
        for on in ("0", "1"):
            os.environ["VGL_GL2X"] = on
            sim = Simulator(VcfglArgs(seed=42, depth=depth, error_rate=0.01, **kw), N, max_sites_per_tile=S, hooks=True)
            sim.timing(True)
            tile = sim.new_tile(S, fields=["fmt_dp", "gl"], device="cuda:0")
            sim.simulate_device(0, gt, tile); sim.check(); sim.kernel_ms(reset=True)
            for _ in range(4):
                sim.simulate_device(0, gt, tile); sim.check()
            ms, n = sim.kernel_ms(reset=True)
            t.append(ms[4] / max(n[4], 1))
            assert sim.info()["gl_wpb"] == (16 if on == "1" else 8)
            sim.close(); del tile
        print(f"{eqs:8d} {N:5d} {depth:5.0f}   {t[0]:10.3f}   {t[1]:10.3f}   {t[1] / t[0]:6.3f}", flush=True)
