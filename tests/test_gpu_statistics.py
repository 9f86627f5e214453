"""Tile mode against serial mode as DISTRIBUTIONS.  VGL_RNG_SERIAL reproduces the reference program draw for draw (golden VCFs);
VGL_RNG_TILE runs the same samplers on other windows of the same generator, so its output differs value by value and must agree
in law.  Same input, same flags, both modes; depth, per-read base errors, quality scores and the
called genotype are compared as histograms (two-sample chi-square) and means.  Seeds are fixed: the test is deterministic."""
import numpy as np
import pytest

import synth
from vcfgl_amd import Simulator, VcfglArgs, _abi

pytestmark = pytest.mark.gpu


def run(args, gt, mode, read_capacity):
    args.rng_mode = mode
    args.beta_sampler = _abi.VGL_BETA_RAND48
    sim = Simulator(args, gt.shape[1], device=0, max_sites_per_tile=gt.shape[0])
    t = sim.simulate(0, gt, read_capacity=read_capacity)
    out = {f: t.numpy(f).copy() for f in ("fmt_dp", "reads", "gl", "fmt_ad")}
    sim.close()
    return out


def chi2_two_sample(a, b):
    """chi-square statistic and degrees of freedom of two count vectors (bins with fewer than 10 expected counts are merged)"""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    keep = (a + b) >= 20
    a2 = np.append(a[keep], a[~keep].sum()); b2 = np.append(b[keep], b[~keep].sum())
    a2, b2 = a2[(a2 + b2) > 0], b2[(a2 + b2) > 0]
    na, nb = a2.sum(), b2.sum()
    stat = (((a2 * np.sqrt(nb / na) - b2 * np.sqrt(na / nb)) ** 2) / (a2 + b2)).sum()
    return stat, len(a2) - 1


@pytest.mark.parametrize("eqs", [0, 2])
def test_tile_mode_agrees_in_distribution_with_the_reference_order(eqs):
    S, N, cap = 1500, 64, 64
    gt = synth.binary_sites(0, S, N)
    kw = dict(error_qs=2, beta_variance=1e-5) if eqs == 2 else {}
    res = {}
    for name, mode in (("serial", _abi.VGL_RNG_SERIAL), ("tile", _abi.VGL_RNG_TILE)):
        res[name] = run(VcfglArgs(seed=123, depth=12.0, error_rate=0.02, add_fmt_ad=1, **kw), gt, mode, cap)
    a, b = res["serial"], res["tile"]
    assert not np.array_equal(a["fmt_dp"], b["fmt_dp"])                     # different windows of the generator
    n = S * N
    # depth: Poisson(12)
    ha, hb = np.bincount(a["fmt_dp"].ravel(), minlength=64), np.bincount(b["fmt_dp"].ravel(), minlength=64)
    stat, dof = chi2_two_sample(ha, hb)
    assert stat < dof + 5 * np.sqrt(2 * dof), ("depth histogram", stat, dof)
    for r in (a, b):
        assert abs(r["fmt_dp"].mean() - 12.0) < 5 * np.sqrt(12.0 / n)
    # per-read base errors: a read's base differs from both alleles of its (binary) genotype with probability e at a homozygote
    # and 2/3 e at a heterozygote (a third of its errors land on the other allele); compare the rates of the two modes and the
    # quality-score histograms
    a0, a1 = (gt & 15)[None], (gt >> 4)[None]

    def read_stats(r):
        rd = r["reads"][:cap]
        depth = r["fmt_dp"][None]
        valid = np.arange(cap)[:, None, None] < depth
        base, q = rd & 3, rd >> 2
        wrong = valid & (base != a0) & (base != a1)
        return wrong.sum(), valid.sum(), np.bincount(q[valid].ravel(), minlength=64)

    wa, na_, qa = read_stats(a)
    wb, nb_, qb = read_stats(b)
    pa, pb = wa / na_, wb / nb_
    se = np.sqrt(pa * (1 - pa) / na_ + pb * (1 - pb) / nb_)
    assert abs(pa - pb) < 5 * se, ("wrong-base rate", pa, pb)
    assert 0.0125 < pa < 0.0205 and 0.0125 < pb < 0.0205                    # e = 0.02 at homozygotes, 2/3 e at heterozygotes
    if eqs == 2:
        stat, dof = chi2_two_sample(qa, qb)
        assert stat < dof + 5 * np.sqrt(2 * dof), ("quality-score histogram", stat, dof)
    else:
        assert np.array_equal(qa > 0, qb > 0)                               # one fixed score
    # called genotype (argmax GL over the three genotypes of the two listed alleles) against the truth: same accuracy
    def accuracy(r):
        gl = r["gl"].reshape(S, -1, N)[:, :3, :]
        ok = np.isfinite(gl).all(axis=1) & (r["fmt_dp"] > 0)
        call = gl.argmax(axis=1)
        truth = (gt & 15).astype(int) + (gt >> 4).astype(int)              # 0, 1, 2 copies of the ALT base (C): genotype index 0 / 1 / 2 when REF is allele 0
        return ok, call, truth
    oka, ca, tr = accuracy(a)
    okb, cb, _ = accuracy(b)
    # allele order may put the ALT first at sites where it is the majority: compare modes only through the rate of calls that
    # equal each other's truth-consistent pattern, i.e. the fraction of heterozygous calls
    ha_, hb_ = (ca[oka] == 1).mean(), (cb[okb] == 1).mean()
    assert abs(ha_ - hb_) < 5 * np.sqrt(ha_ * (1 - ha_) / oka.sum() * 2), ("heterozygous call rate", ha_, hb_)
