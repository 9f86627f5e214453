"""Pins the CPU oracle (oracle/vgl_oracle.c, reference stream order) against the
reference's own golden VCFs (test/reference/*, copied under tests/golden/ref_vcf)."""
import os

import numpy as np
import pytest

import golden_util as gu
from vcfgl_amd import _abi

CASES = sorted((k for k, v in gu.REF_TESTS.items() if not v.get("cli_only")), key=lambda s: int(s[4:]))


@pytest.mark.parametrize("name", CASES)
def test_reference_golden_vcf(oracle, name):
    args, vcf, sites, gold = gu.load_case(name)
    o = oracle.Oracle(args, len(vcf.samples))
    gt = np.stack([s.gt for s in sites])
    tile = o.simulate(0, gt, read_capacity=0)
    errs = gu.compare_with_golden(args, sites, tile, gold)
    assert not errs, "\n".join(errs[:40])


def test_reference_golden_pileup(oracle):
    """test10's pileup pins every simulated read's base and quality score."""
    args, vcf, sites, gold = gu.load_case("test10")
    o = oracle.Oracle(args, len(vcf.samples))
    gt = np.stack([s.gt for s in sites])
    cap = 16
    tile = o.simulate(0, gt, read_capacity=cap)
    rows = gu.read_pileup(os.path.join(gu.REFVCF, "reference", "test10", "test10.pileup.gz"))
    status = tile.numpy("site_status")
    reads = tile.numpy("reads")
    dp = tile.numpy("fmt_dp")
    assert len(rows) == len(sites)
    for i, (chrom, pos, ref, smp) in enumerate(rows):
        assert (chrom, pos) == (sites[i].chrom, sites[i].pos0 + 1)
        for s, (n, bases, quals) in enumerate(smp):
            assert n == dp[i, s]
            if n == 0:
                assert bases == "*" and quals == "*"
                continue
            ours_b = "".join("ACGT"[reads[r, i, s] & 3] for r in range(n))
            ours_q = "".join(chr((reads[r, i, s] >> 2) + 33) for r in range(n))
            assert (ours_b, ours_q) == (bases, quals), (chrom, pos, s)


DOC_RUNS = {
    # documented runs of the reference (doc/with_msprime.MD, doc/error_qs.MD): flags, input, expected VCF listing
    "msprime": ("--depth 10 --error-rate 0 --source 0 --seed 42", "doc_msprime/msprime_output.vcf", "doc_msprime/sim_source0.vcf"),
    "error_qs0": ("--depth 2 --error-rate 0.4 --error-qs 0 -addFormatAD 1 --seed 42", "ref_vcf/data/data2.vcf", "doc_error_qs/error_qs0.vcf"),
    "error_qs1": ("--depth 2 --error-rate 0.4 --error-qs 1 --beta-variance 1e-1 -addFormatAD 1 --seed 42", "ref_vcf/data/data2.vcf", "doc_error_qs/error_qs1.vcf"),
    "error_qs2": ("--depth 2 --error-rate 0.4 --error-qs 2 --beta-variance 1e-1 -addFormatAD 1 --seed 42", "ref_vcf/data/data2.vcf", "doc_error_qs/error_qs2.vcf"),
}


@pytest.mark.parametrize("name", sorted(DOC_RUNS))
def test_documented_runs_of_the_reference(oracle, name):
    """The output listings in the reference's documentation pin the oracle beyond its test-suite: depth 10 (eleven
    to sixteen reads per sample, quality score capped at 63) and the three --error-qs modes at error rate 0.4."""
    from vcfgl_amd.params import VcfglArgs
    from vcfgl_amd.recordloop import iter_sites
    from vcfgl_amd.vcfio import read_vcf
    flags, inp, exp = DOC_RUNS[name]
    args = VcfglArgs.from_argv(flags.split()).validate()
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_SERIAL, _abi.VGL_BETA_STD
    vcf = read_vcf(os.path.join(gu.GOLD, inp))
    sites = list(iter_sites(vcf, args))
    gold = read_vcf(os.path.join(gu.GOLD, exp))
    tile = oracle.Oracle(args, len(vcf.samples)).simulate(0, np.stack([s.gt for s in sites]))
    errs = gu.compare_with_golden(args, sites, tile, gold)
    assert not errs, "\n".join(errs[:40])
