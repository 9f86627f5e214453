"""Diagnostic: does a second (third) context on its own stream overlap its tiles with the first's?  usage (GPU box): python tools/two_streams.py [c3|fixedq|alltags|c5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from vcfgl_amd import Simulator, VcfglArgs, _abi
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
if wl == "c3":
    N, S, T = 1000, 65536, 16
    a = dict(seed=42, depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5)
    fields = ["fmt_dp", "gl"]
elif wl == "fixedq":
    N, S, T = 1000, 65536, 16
    a = dict(seed=42, depth=20.0, error_rate=0.01)
    fields = ["fmt_dp", "gl"]
elif wl == "alltags":
    N, S, T = 1000, 65536, 4
    a = dict(seed=42, depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, add_gp=1, add_pl=1, add_qs=1, add_i16=1, add_info_dp=1, add_fmt_ad=1, add_info_ad=1, add_fmt_adf=1, add_info_adf=1, add_fmt_adr=1, add_info_adr=1)
    fields = None
else:
    N, S, T = 500, 65536, 32
    a = dict(seed=42, depth=5.0, error_rate=0.01, explode=1, do_unobserved=2, add_pl=1)
    fields = ["fmt_dp", "gl", "pl"]
gt = synth.binary_sites_torch(0, S, N, "cuda:0") if wl != "c5" else torch.zeros((S, N), dtype=torch.uint8, device="cuda:0")
for nctx in (1, 2, 3, 1, 2):
    sims, tiles, streams = [], [], []
    for k in range(nctx):
        args = VcfglArgs(**a); args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
        sim = Simulator(args, N, max_sites_per_tile=S)
        sims.append(sim); tiles.append(sim.new_tile(S, fields=fields or sim.default_fields(), device="cuda:0")); streams.append(torch.cuda.Stream())
    def step():
        for t in range(T):
            k = t % nctx
            sims[k].simulate_device(t * S, gt, tiles[k], stream=streams[k].cuda_stream)
    step(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 3
    for sim in sims: sim.check(); sim.close()
    print(f"{wl}: {nctx} context(s)/stream(s): {dt * 1e3:.1f} ms per {T} tiles = {T * S * N / dt:.4e} evaluations/s", flush=True)
    del tiles; torch.cuda.empty_cache()
