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
