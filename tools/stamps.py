#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of k_sample<2> (in-kernel clock64 stamps) and, for the float32 build, what its pool loop does per
iteration -- how often each rarely-run block of the loop executes and how many lanes do useful work (the block weights tools/isa_hist.py
takes: VERDICT r5 item 1a).  Shares and counts only -- a stamped run is never a timing result.

usage (GPU box): python tools/stamps.py [1|2] [depth] [json out]     1 = the inline-fallback build (LEAN 0), 2 = the float32 build of the default tag surface (LEAN 2)"""
import ctypes as C, json, os, sys
mode = sys.argv[1] if len(sys.argv) > 1 else "1"
depth = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
os.environ["VGL_DEBUG_STAMPS"] = mode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from vcfgl_amd import Simulator, VcfglArgs, _abi
N, S = 1000, 8192
a = VcfglArgs(seed=42, depth=depth, error_rate=0.01, error_qs=2, beta_variance=1e-5)
a.rng_mode, a.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
sim = Simulator(a, N, max_sites_per_tile=S, hooks=True)
gt = synth.binary_sites_torch(0, S, N, "cuda:0")
tile = sim.new_tile(S, fields=["fmt_dp", "gl"], device="cuda:0")
sim.simulate_device(0, gt, tile); sim.check()
out = (C.c_ulonglong * 16)()
sim.lib.vgl_dbg_stamps.argtypes = [C.c_void_p, C.c_void_p]
assert sim.lib.vgl_dbg_stamps(sim.ctx, out) == 0
w = out[0]
names = ["waves", "cycles/wave", "streams+poisson", "owner(bases)", "pool(beta)", "flush", "pool iterations", "pool items",
         "normal log-test blocks", "gamma log-test blocks", "finish blocks", "lanes with an item", "lanes finishing", "lanes holding"]
res = {}
for n, v in zip(names, out):
    res[n] = v if n == "waves" else v / w
    print(f"{n:24s} {v / w:12.2f}" if n != "waves" else f"{n:24s} {v}")
if mode == "2" and res["pool iterations"] > 0:
    it = res["pool iterations"]
    res["per_iteration"] = {"normal_test": res["normal log-test blocks"] / it, "gamma_test": res["gamma log-test blocks"] / it, "finish": res["finish blocks"] / it,
                            "lanes_with_item": res["lanes with an item"] / it, "lanes_finishing": res["lanes finishing"] / it, "lanes_holding": res["lanes holding"] / it,
                            "iterations_per_item_at_64_lanes": 64.0 * it / max(res["pool items"], 1)}
    print("per iteration:", json.dumps(res["per_iteration"]))
    print(f"ideal iterations (items x lane-iterations per item / 64) vs actual: lane efficiency of the loop = "
          f"{(res['lanes with an item'] - res['lanes holding']) / (64.0 * it):.3f} (lanes with an item and not holding / 64)")
if len(sys.argv) > 3:
    json.dump(res, open(sys.argv[3], "w"), indent=1)
