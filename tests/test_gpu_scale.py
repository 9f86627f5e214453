"""Size-independent properties at (a slice of) the benchmark's size: the device-resident path of
bench.py (config C3 flags, 1000 samples) checked through invariants that do not need the oracle to
run the whole job: tiling invariance of a checksum of checksums, determinism, oracle equality on
sampled sites, Poisson/accounting identities, allele-order invariants."""
import ctypes as C

import numpy as np
import pytest
import torch

import synth
from vcfgl_amd import Simulator, VcfglArgs, _abi

pytestmark = pytest.mark.gpu
N = 1000
S = 24576            # 2.4e7 evaluations, 3 bench-sized tiles


def _args(**kw):
    a = VcfglArgs(seed=42, depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, add_fmt_ad=1, add_info_ad=1, **kw)
    a.rng_mode, a.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    return a


def _run(args, tile_sites, site0=0, n_sites=S):
    dev = torch.device("cuda", 0)
    sim = Simulator(args, N, max_sites_per_tile=tile_sites)
    gt = synth.binary_sites_torch(site0, n_sites, N, dev)
    out = {"site_status": torch.empty((n_sites,), dtype=torch.int32, device=dev),
           "n_alleles": torch.empty((n_sites,), dtype=torch.int32, device=dev),
           "alleles2acgt": torch.empty((n_sites, 5), dtype=torch.int8, device=dev),
           "info_ad": torch.empty((n_sites, sim.A), dtype=torch.int32, device=dev),
           "fmt_dp": torch.empty((n_sites, N), dtype=torch.int32, device=dev),
           "fmt_ad": torch.empty((n_sites, sim.A, N), dtype=torch.int32, device=dev),
           "gl": torch.empty((n_sites, sim.G, N), dtype=torch.float32, device=dev)}
    for s0 in range(0, n_sites, tile_sites):
        n = min(tile_sites, n_sites - s0)
        t = _abi.TileOut()
        for k, v in out.items():
            setattr(t, k, v[s0:s0 + n].data_ptr())
        sim._check(sim.lib.vgl_simulate_tile_device(sim.ctx, site0 + s0, n, gt[s0:s0 + n].data_ptr(), C.byref(t), None))
    sim.check()
    sim.close()
    return gt, out


def _checksum(out):
    """checksum of per-array checksums (int64 wrap-around sums of the raw bits)"""
    tot = 0
    for k in sorted(out):
        v = out[k]
        bits = v.view(torch.int32) if v.dtype == torch.float32 else v
        tot = (tot * 1000003 + int(bits.to(torch.int64).sum().item())) % (1 << 61)
    return tot


def test_tiling_invariance_and_determinism():
    a = _args()
    _, o1 = _run(a, 8192)
    _, o2 = _run(a, 5000)          # ragged last tile
    _, o3 = _run(a, 8192)
    c1, c2, c3 = _checksum(o1), _checksum(o2), _checksum(o3)
    assert c1 == c2 == c3
    for k in o1:
        assert torch.equal(o1[k].view(torch.int32) if o1[k].dtype == torch.float32 else o1[k],
                           o2[k].view(torch.int32) if o2[k].dtype == torch.float32 else o2[k]), k


def test_accounting_identities_and_statistics():
    a = _args()
    gt, o = _run(a, 8192)
    dp, ad, gl = o["fmt_dp"], o["fmt_ad"], o["gl"]
    assert torch.equal(ad.sum(dim=1), dp)                                   # every read is one allele count
    assert torch.equal(o["info_ad"], ad.sum(dim=2))                         # INFO/AD = sum of FORMAT/AD
    ia = o["info_ad"][:, :4]
    nobs = (o["n_alleles"] - 1).clamp(min=0)
    for k in range(3):                                                      # alleles sorted by depth, descending
        valid = nobs > k + 1
        assert bool((ia[valid, k] >= ia[valid, k + 1]).all())
    mean = dp.double().mean().item()
    assert abs(mean - 20.0) < 0.01, mean                                    # Poisson(20) over 2.4e7 draws
    var = dp.double().var().item()
    assert abs(var - 20.0) < 0.05, var
    # GL: per sample the best genotype is exactly 0 and nothing is positive (gl_methods.cpp:50-58)
    bits = gl.view(torch.int32)
    miss = bits == 0x7F800001
    g = torch.where(miss, torch.full_like(gl, -1.0), gl)
    assert float(g.max()) == 0.0
    has = dp > 0
    best = torch.where(miss, torch.full_like(gl, float("-inf")), gl).max(dim=1).values
    assert bool((best[has] == 0.0).all())
    assert bool(miss.all(dim=1)[~has].all())                               # DP 0 <=> all genotypes missing
    # observed error rate: reads that match neither true allele, hom-ref samples
    hom_ref = gt == 0
    a2b = o["alleles2acgt"]
    refcol = (a2b == 0).to(torch.int64).argmax(dim=1)                       # allele index of base A per site
    ad_a = torch.gather(ad, 1, refcol.view(-1, 1, 1).expand(-1, 1, N)).squeeze(1)
    has_a = (a2b == 0).any(dim=1).view(-1, 1)
    wrong = ((dp - ad_a) * (hom_ref & has_a)).sum().item() / max(1, (dp * (hom_ref & has_a)).sum().item())
    assert abs(wrong - 0.01) < 2e-4, wrong


def test_sampled_sites_equal_oracle(oracle):
    """oracle equality on sites scattered through the job (absolute site indexing)"""
    a = _args()
    _, o = _run(a, 8192)
    orc = oracle.Oracle(a, N)
    for site in (0, 1, 4097, 8191, 8192, 20000, S - 1):
        want = orc.simulate(site, synth.binary_sites(site, 1, N), fields=["fmt_dp", "gl", "fmt_ad", "info_ad"])
        assert np.array_equal(o["fmt_dp"][site].cpu().numpy(), want.numpy("fmt_dp")[0]), site
        assert np.array_equal(o["fmt_ad"][site].cpu().numpy(), want.numpy("fmt_ad")[0]), site
        assert np.array_equal(o["gl"][site].cpu().numpy().view(np.uint32), want.numpy("gl")[0].view(np.uint32)), site
        assert np.array_equal(o["alleles2acgt"][site].cpu().numpy(), want.numpy("alleles2acgt")[0]), site


def test_full_c3_job_tiling_invariance_and_spot_parity(oracle):
    """BASELINE.json configs[2] at its full size -- 1,000,000 sites x 1000 samples, depth 20, --error-qs 2, GL 2 (1e9
    evaluations, 2e10 reads), the job bench.py times -- through size-independent properties: a checksum of per-tile
    checksums that two different tilings must agree on, the accounting identity DP == sum(AD) on every tile, the Poisson
    mean over all 1e9 depth draws, and oracle equality on sites scattered up to the last one."""
    S_FULL = 1_000_000
    dev = torch.device("cuda", 0)
    a = _args()
    keep = (0, 16383, 16384, 500_000, 777_777, S_FULL - 1)

    def job(tile_sites):
        sim = Simulator(a, N, max_sites_per_tile=tile_sites)
        buf = {"site_status": torch.empty((tile_sites,), dtype=torch.int32, device=dev),
               "n_alleles": torch.empty((tile_sites,), dtype=torch.int32, device=dev),
               "alleles2acgt": torch.empty((tile_sites, 5), dtype=torch.int8, device=dev),
               "fmt_dp": torch.empty((tile_sites, N), dtype=torch.int32, device=dev),
               "fmt_ad": torch.empty((tile_sites, sim.A, N), dtype=torch.int32, device=dev),
               "gl": torch.empty((tile_sites, sim.G, N), dtype=torch.float32, device=dev)}
        sums = {k: torch.zeros((), dtype=torch.int64, device=dev) for k in buf}
        dp_total = torch.zeros((), dtype=torch.int64, device=dev)
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        kept = {}
        for s0 in range(0, S_FULL, tile_sites):
            n = min(tile_sites, S_FULL - s0)
            gt = synth.binary_sites_torch(s0, n, N, dev)
            t = _abi.TileOut()
            for k, v in buf.items():
                setattr(t, k, v.data_ptr())
            sim._check(sim.lib.vgl_simulate_tile_device(sim.ctx, s0, n, gt.data_ptr(), C.byref(t), None))
            for k, v in buf.items():                               # wrap-around int64 sums of the raw bits
                x = v[:n]
                sums[k] += (x.view(torch.int32) if x.dtype == torch.float32 else x).sum(dtype=torch.int64)
            dp_total += buf["fmt_dp"][:n].sum(dtype=torch.int64)
            bad += (buf["fmt_ad"][:n].sum(dim=1) != buf["fmt_dp"][:n]).sum()
            for site in keep:
                if s0 <= site < s0 + n:
                    kept[site] = {k: buf[k][site - s0].cpu().numpy().copy() for k in ("fmt_dp", "fmt_ad", "gl", "alleles2acgt")}
        sim.check()
        sim.close()
        tot = 0
        for k in sorted(sums):
            tot = (tot * 1000003 + int(sums[k].item())) % (1 << 61)
        return tot, int(dp_total.item()), int(bad.item()), kept

    c1, dp1, bad1, kept1 = job(16384)
    c2, dp2, bad2, kept2 = job(10000)
    assert c1 == c2 and dp1 == dp2
    assert bad1 == 0 and bad2 == 0
    assert abs(dp1 / (S_FULL * N) - 20.0) < 2e-3                   # sd of the mean of 1e9 Poisson(20) draws: 1.4e-4
    orc = oracle.Oracle(a, N)
    for site in keep:
        want = orc.simulate(site, synth.binary_sites(site, 1, N), fields=["fmt_dp", "gl", "fmt_ad"])
        for got in (kept1[site], kept2[site]):
            assert np.array_equal(got["fmt_dp"], want.numpy("fmt_dp")[0]), site
            assert np.array_equal(got["fmt_ad"], want.numpy("fmt_ad")[0]), site
            assert np.array_equal(got["gl"].view(np.uint32), want.numpy("gl")[0].view(np.uint32)), site
            assert np.array_equal(got["alleles2acgt"], want.numpy("alleles2acgt")[0]), site
