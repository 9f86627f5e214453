"""Diagnostic (round 6): what k_sample<0> pays for -- the base-call error branch, the spread of the 64 depths of a wavefront, heterozygous evaluations.
Per-kernel time of one 65536-site x 1000-sample tile, one fixed score, GL 2, for error rates 0.01 / 0.001 / 1e-6 at depth 20, for depths 12 / 40,
and for an all-homozygous tile.   usage (GPU box): python tools/r6_fixedq_probe.py [VGL_LIB=...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from vcfgl_amd import Simulator, VcfglArgs, _abi
N, S = 1000, 65536
gt_mix = synth.binary_sites_torch(0, S, N, "cuda:0")
gt_hom = torch.zeros_like(gt_mix)
for name, kw, gt in (("e 0.01 depth 20", dict(depth=20.0, error_rate=0.01), gt_mix), ("e 0.001", dict(depth=20.0, error_rate=0.001), gt_mix),
                     ("e 1e-6", dict(depth=20.0, error_rate=1e-6), gt_mix), ("e 0.01 all hom-ref", dict(depth=20.0, error_rate=0.01), gt_hom),
                     ("e 0.01 depth 12", dict(depth=12.0, error_rate=0.01), gt_mix), ("e 0.01 depth 40", dict(depth=40.0, error_rate=0.01), gt_mix)):
    a = VcfglArgs(seed=42, gl_model=2, **kw)
    a.rng_mode, a.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    sim = Simulator(a, N, max_sites_per_tile=S)
    sim.timing(True)
    tile = sim.new_tile(S, fields=["fmt_dp", "gl"], device="cuda:0")
    sim.simulate_device(0, gt, tile); sim.check(); sim.kernel_ms(reset=True)
    for _ in range(3):
        sim.simulate_device(0, gt, tile); sim.check()
    ms, n = sim.kernel_ms(reset=True)
    print(f"{name:22s}", [round(x / max(k, 1), 3) for x, k in zip(ms, n)], "ms per tile (depth, sample, redo, site, gl, siteagg)", flush=True)
    sim.close()
    del tile
