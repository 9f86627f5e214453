#!/usr/bin/env python3
"""Diagnostic: re-run one configuration of tests/test_gpu_fuzz.py (seed base, chunk, index) and print where the
HIP path and the oracle differ.   python tools/fuzz_repro.py 50000 74 7"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as oracle
import test_gpu_fuzz as tf
from vcfgl_amd import Simulator, VcfglArgs, VcfglArgError, _abi

base_seed, chunk, want_idx = (int(x) for x in sys.argv[1:4])
rng = np.random.default_rng(base_seed + chunk)
done = 0
while True:
    kw, N, gt = tf.random_case(rng)
    try:
        VcfglArgs(**kw).validate()
    except VcfglArgError:
        continue
    if done == want_idx:
        break
    done += 1
print(kw, N, gt.shape)
for mode, beta in ((_abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48), (_abi.VGL_RNG_SERIAL, _abi.VGL_BETA_STD)):
    args = VcfglArgs(**kw); args.rng_mode, args.beta_sampler = mode, beta
    orc = oracle.Oracle(args, N)
    sim = Simulator(args, N, max_sites_per_tile=gt.shape[0])
    fields = sim.default_fields()
    want = orc.simulate(0, gt, fields=fields, read_capacity=64)
    got = sim.simulate(0, gt, fields=fields, read_capacity=64)
    sim.close()
    for f in fields + ["reads"]:
        a, b = want.numpy(f), got.numpy(f)
        if a.dtype == np.float32: a, b = a.view(np.uint32), b.view(np.uint32)
        d = np.argwhere(a != b)
        if len(d):
            print("mode", mode, f, len(d), "diffs; first:", d[:6].tolist())
            for idx in d[:6]:
                idx = tuple(idx)
                print("   ", idx, want.numpy(f)[idx], got.numpy(f)[idx])
            if f in ("gl", "pl"):
                i, g, s = tuple(d[0])
                print("    site", i, "sample", s, "dp", want.numpy("fmt_dp")[i, s], "n_alleles", want.numpy("n_alleles")[i], "a2b", want.numpy("alleles2acgt")[i])
                print("    want gl", want.numpy("gl")[i, :, s]); print("    got  gl", got.numpy("gl")[i, :, s])
                print("    want pl", want.numpy("pl")[i, :, s]); print("    got  pl", got.numpy("pl")[i, :, s])
                n = want.numpy("fmt_dp")[i, s]
                print("    reads want", want.numpy("reads")[:n, i, s], "got", got.numpy("reads")[:n, i, s])
