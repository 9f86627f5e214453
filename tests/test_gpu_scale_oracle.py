"""Oracle parity at scale: >= 1e8 evaluations of the benchmark configurations compared with the CPU oracle FIELD BY FIELD
and SITE BY SITE (one 64-bit checksum per site and field on both sides, tests/oracle_pool.py), not by sampling.

  C3  BASELINE.json configs[2] flags, 100,000 sites x 1000 samples = 1e8 evaluations (2e9 reads)
  C4  configs[3] flags (depth 30, rta3 qs-bins), 5,000 sites x 2000 samples = 1e7 evaluations
  C5  configs[4] flags (exploded hom-ref sites, -doUnobserved 2, PL), 200,000 sites x 500 samples = 1e8 evaluations

The site ranges start far from 0 (absolute site indexing: the same values the full job produces there).  Integer
fields and GL are bit-exact on these paths (GL terms come from the qScore LUT), so the checksums must be EQUAL."""
import ctypes as C
import dataclasses

import numpy as np
import pytest
import torch

import oracle_pool
import synth
from vcfgl_amd import Simulator, VcfglArgs, _abi

pytestmark = pytest.mark.gpu

RTA3 = [(0, 2, 2), (3, 14, 12), (15, 30, 23), (31, 40, 37)]
CASES = {
    "c3": dict(N=1000, S=100_000, site0=400_000, gt="binary", fields=["site_status", "n_alleles", "alleles2acgt", "fmt_dp", "gl"],
               flags=dict(depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=2)),
    "c4": dict(N=2000, S=5_000, site0=9_000_000, gt="binary", fields=["site_status", "n_alleles", "alleles2acgt", "fmt_dp", "gl"],
               flags=dict(depth=30.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=2, qs_bins=RTA3)),
    "c5": dict(N=500, S=200_000, site0=40_000_000, gt="homref", fields=["site_status", "n_alleles", "alleles2acgt", "fmt_dp", "gl", "pl"],
               flags=dict(depth=5.0, error_rate=0.01, gl_model=2, do_unobserved=2, add_pl=1)),
}
TORCH_DT = {"int32": torch.int32, "int8": torch.int8, "float32": torch.float32}


def _args(flags):
    a = VcfglArgs(seed=42, **flags)
    a.rng_mode, a.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    return a


def _gpu_checksums(case, tile_sites=8192):
    dev = torch.device("cuda", 0)
    a = _args(case["flags"])
    N, S, site0 = case["N"], case["S"], case["site0"]
    sim = Simulator(a, N, max_sites_per_tile=tile_sites)
    shapes = {"site": (), "site5": (5,), "eval": (N,), "planeG": (sim.G, N)}
    kinds = {f: (dt, k) for f, dt, k in _abi.TILE_FIELDS}
    buf = {f: torch.empty((tile_sites,) + shapes[kinds[f][1]], dtype=TORCH_DT[kinds[f][0]], device=dev) for f in case["fields"]}
    out = np.zeros((S, len(case["fields"])), dtype=np.uint64)
    for s0 in range(0, S, tile_sites):
        n = min(tile_sites, S - s0)
        gt = torch.zeros((n, N), dtype=torch.uint8, device=dev) if case["gt"] == "homref" else synth.binary_sites_torch(site0 + s0, n, N, dev)
        t = _abi.TileOut()
        for f, v in buf.items():
            setattr(t, f, v.data_ptr())
        sim._check(sim.lib.vgl_simulate_tile_device(sim.ctx, site0 + s0, n, gt.data_ptr(), C.byref(t), None))
        sim.check()
        for k, f in enumerate(case["fields"]):
            out[s0:s0 + n, k] = oracle_pool.site_checksums_torch(buf[f][:n])
    sim.close()
    return out


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["c3", "c4", "c5"])
def test_every_site_equals_oracle(oracle, name):
    case = CASES[name]
    got = _gpu_checksums(case)
    d = dataclasses.asdict(_args(case["flags"]))
    want = oracle_pool.oracle_site_checksums(d, case["N"], case["site0"], case["S"], case["fields"], gt=case["gt"])
    assert got.shape == want.shape == (case["S"], len(case["fields"]))
    bad = np.argwhere(got != want)
    assert bad.size == 0, (f"{name}: {len(bad)} (site, field) checksums differ; first: site {case['site0'] + bad[0][0]} "
                           f"field {case['fields'][bad[0][1]]}")
    print(f"{name}: {case['S'] * case['N']:.3g} evaluations, {case['S']} sites x {len(case['fields'])} fields equal to the oracle")
