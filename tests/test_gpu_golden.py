"""The HIP path itself, in VGL_RNG_SERIAL mode (sequential scout + parallel kernels), run on the
reference's own test inputs with the reference's flags, against the reference's golden VCFs
(tests/golden/ref_vcf = data files of /root/reference/test).  I16 fields 13-16 (tail distance,
the reference's unseeded libc rand()) are reproduced too: the scout restates glibc's generator."""
import os

import numpy as np
import pytest

import golden_util as gu
from vcfgl_amd import Simulator, _abi

pytestmark = pytest.mark.gpu
CASES = sorted((k for k, v in gu.REF_TESTS.items() if not v.get("cli_only")), key=lambda s: int(s[4:]))


@pytest.mark.parametrize("name", CASES)
def test_device_reproduces_reference_golden_vcf(name):
    args, vcf, sites, gold = gu.load_case(name, rng_mode=_abi.VGL_RNG_SERIAL, beta_sampler=_abi.VGL_BETA_STD)
    gt = np.stack([s.gt for s in sites])
    sim = Simulator(args, len(vcf.samples), max_sites_per_tile=len(sites))
    tile = sim.simulate(0, gt)
    sim.close()
    errs = gu.compare_with_golden(args, sites, tile, gold, check_i16_tail=True)
    assert not errs, "\n".join(errs[:40])


def test_device_reproduces_reference_pileup():
    args, vcf, sites, gold = gu.load_case("test10", rng_mode=_abi.VGL_RNG_SERIAL, beta_sampler=_abi.VGL_BETA_STD)
    gt = np.stack([s.gt for s in sites])
    sim = Simulator(args, len(vcf.samples), max_sites_per_tile=len(sites))
    tile = sim.simulate(0, gt, read_capacity=16)
    sim.close()
    rows = gu.read_pileup(os.path.join(gu.REFVCF, "reference", "test10", "test10.pileup.gz"))
    reads, dp = tile.numpy("reads"), tile.numpy("fmt_dp")
    for i, (chrom, pos, ref, smp) in enumerate(rows):
        for s, (n, bases, quals) in enumerate(smp):
            assert n == dp[i, s]
            if n:
                assert "".join("ACGT"[reads[r, i, s] & 3] for r in range(n)) == bases
                assert "".join(chr((reads[r, i, s] >> 2) + 33) for r in range(n)) == quals


def test_serial_streams_continue_across_tiles(oracle):
    """two consecutive tiles == one tile: the generator states are carried in the context;
    also equals the CPU oracle's serial mode on a larger random input"""
    import synth
    from vcfgl_amd import VcfglArgs, VglError
    args = VcfglArgs(seed=7, depth=3, error_rate=0.02, error_qs=2, beta_variance=1e-5, add_pl=1, add_fmt_ad=1, add_qs=1)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_SERIAL, _abi.VGL_BETA_STD
    N, S = 37, 50
    gt = synth.acgt_sites(S, N, seed=5, missing=0.05)
    sim = Simulator(args, N, max_sites_per_tile=S)
    a = sim.simulate(0, gt[:20])
    b = sim.simulate(20, gt[20:])
    with pytest.raises(VglError):
        sim.simulate(0, gt[:5])                      # out of order: refused, not silently different
    sim.close()
    want = oracle.Oracle(args, N).simulate(0, gt, fields=sim.default_fields())
    for f in ["fmt_dp", "fmt_ad", "pl", "alleles2acgt", "site_status"]:
        assert np.array_equal(np.concatenate([a.numpy(f), b.numpy(f)]), want.numpy(f)), f
    got_gl = np.concatenate([a.numpy("gl"), b.numpy("gl")]).view(np.uint32)
    assert np.array_equal(got_gl, want.numpy("gl").view(np.uint32))
    assert np.array_equal(np.concatenate([a.numpy("qs"), b.numpy("qs")]).view(np.uint32), want.numpy("qs").view(np.uint32))


@pytest.mark.parametrize("kw", [
    dict(depth=20, error_rate=0.01),                                              # rejection Poisson, no strand
    dict(depth=30, error_rate=0.05, add_i16=1, add_fmt_adf=1, add_fmt_adr=1),     # strand draws + I16 (libc rand)
    dict(depth=5, error_rate=0.2),                                                # product-method Poisson, many errors
    dict(depth=14, error_rate=0.03, error_qs=1, beta_variance=1e-4),              # per-site beta error rate
    dict(depths=[0.0, 25.0, 3.0, 12.0] * 50, error_rate=0.02, add_i16=1),         # per-sample means incl. zero
    dict(depth=100, error_rate=0.0, do_unobserved=5),                             # blocks of > 64 reads per sample
])
def test_wave_scout_equals_oracle_serial(oracle, kw):
    """the wave-parallel state scout (no per-read beta) against the CPU oracle's serial mode:
    every integer field, GL bits, QS bits, I16 incl. tail distances"""
    import synth
    from vcfgl_amd import VcfglArgs
    args = VcfglArgs(seed=42, add_pl=1, add_qs=1, add_fmt_ad=1, add_info_ad=1, add_info_dp=1, **kw)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_SERIAL, _abi.VGL_BETA_STD
    N, S = 200, 30
    gt = synth.acgt_sites(S, N, seed=11, missing=0.03)
    sim = Simulator(args, N, max_sites_per_tile=16)
    a = sim.simulate(0, gt[:16])
    b = sim.simulate(16, gt[16:])
    fields = sim.default_fields()
    sim.close()
    want = oracle.Oracle(args, N).simulate(0, gt, fields=fields)
    for f in ["site_status", "alleles2acgt", "n_alleles", "info_dp", "info_ad", "info_adf", "info_adr", "fmt_dp", "fmt_ad", "fmt_adf", "fmt_adr", "pl"]:
        assert np.array_equal(np.concatenate([a.numpy(f), b.numpy(f)]), want.numpy(f)), f
    for f in ["gl", "qs"] + (["i16"] if args.add_i16 else []):
        got = np.concatenate([a.numpy(f), b.numpy(f)])
        assert np.array_equal(got.view(np.uint32), want.numpy(f).view(np.uint32)), f
