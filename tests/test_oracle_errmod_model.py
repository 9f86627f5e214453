"""GL model 1 beyond the depth the reference's golden files reach (n <= 3): the oracle's restatement of htslib's errmod
(oracle/vgl_oracle.c errmod_make / errmod_cal5) against an INDEPENDENT evaluation of the published model ("revised MAQ",
Li 2011) written here from its definition in exact rational arithmetic -- not from errmod.c's
recurrences:

    beta(q, n, k) = -10 log10( P(X >= k+1) / P(X >= k) ),   X ~ Binomial(n, e),  e = 10^(-q/10)
    fk(w)         = (1 - depcorr)^w (1 - eta) + eta,  eta = 0.03
    reads in descending (quality, base) order; per base b with running count c_b and per (strand, base) count w:
        bsum_b += fk(w) beta(q, n, c_b)
    homozygous j:      sum_{b != j} bsum_b                                   (0 when no other base was seen)
    heterozygous j,k:  -4.343 (ln C(c_j + c_k, c_k) - (c_j + c_k) ln 2) + sum_{b not in {j,k}} bsum_b

This pins the MODEL at depth up to 255; what stays unpinned is htslib's own floating-point evaluation order (its source is
not in the reference tree), which can move the float32 results in their last bits only."""
import ctypes as C
import math
from fractions import Fraction

import numpy as np
import pytest


def binom_tail_ratios(n, e):
    """[P(X >= k+1) / P(X >= k) for k = 0..n-1] exactly, X ~ Binomial(n, e), e a Fraction"""
    pmf = [Fraction(math.comb(n, j)) * e ** j * (1 - e) ** (n - j) for j in range(n + 1)]
    tail = [Fraction(0)] * (n + 2)
    for j in range(n, -1, -1):
        tail[j] = tail[j + 1] + pmf[j]
    return [tail[k + 1] / tail[k] for k in range(n)]


def model_errmod(depcorr, codes):
    """codes: qual << 5 | base (strand 0).  Returns the 5x5 matrix of phred-scaled costs (float64)."""
    n_all = len(codes)
    assert n_all <= 255
    eta = 0.03
    fk = lambda w: (1.0 - depcorr) ** w * (1.0 - eta) + eta
    order = sorted(codes, reverse=True)
    c, w, bsum = [0] * 5, [0] * 5, [0.0] * 5
    cache = {}
    for code in order:
        q = min(max(code >> 5, 4), 63)
        b = code & 0xF
        if q not in cache:
            e = Fraction(10.0 ** (-q / 10.0))
            cache[q] = [-10.0 * math.log10(float(r)) for r in binom_tail_ratios(n_all, e)]
        bsum[b] += fk(w[b]) * cache[q][c[b]]
        c[b] += 1; w[b] += 1
    out = np.zeros((5, 5))
    for j in range(5):
        others = [k for k in range(5) if k != j]
        out[j, j] = sum(bsum[k] for k in others) if sum(c[k] for k in others) else 0.0
        for k in range(j + 1, 5):
            rest = [i for i in range(5) if i not in (j, k)]
            cjk = c[j] + c[k]
            lhet = math.log(math.comb(cjk, c[k])) - cjk * math.log(2.0)
            v = -4.343 * lhet + (sum(bsum[i] for i in rest) if sum(c[i] for i in rest) else 0.0)
            out[j, k] = out[k, j] = max(v, 0.0)
        out[j, j] = max(out[j, j], 0.0)
    return out


def oracle_errmod(oracle, depcorr, codes):
    lib = oracle.lib()
    arr = (C.c_uint16 * len(codes))(*codes)
    q = (C.c_float * 25)()
    assert lib.vgl_oracle_errmod_cal(depcorr, len(codes), arr, q) == 0
    return np.array(list(q), dtype=np.float64).reshape(5, 5)


@pytest.mark.parametrize("n", [1, 2, 3, 7, 40, 131, 255])
def test_errmod_restatement_equals_the_model_definition(oracle, n):
    rng = np.random.default_rng(100 + n)
    for trial in range(3):
        major = int(rng.integers(0, 4))
        bases = np.where(rng.random(n) < 0.8, major, rng.integers(0, 4, n))
        quals = rng.choice([2, 7, 20, 37, 63], size=n) if trial else np.full(n, 20)
        codes = [int(q) << 5 | int(b) for q, b in zip(quals, bases)]
        for depcorr in (0.17, 0.0):
            got, want = oracle_errmod(oracle, depcorr, codes), model_errmod(depcorr, codes)
            assert np.allclose(got, want, rtol=2e-6, atol=1e-4), (n, trial, depcorr, np.abs(got - want).max())


def test_known_answer_telescoping_sum(oracle):
    """depcorr 0 makes every fk = 1; n reads of one base at quality q then cost any other homozygote sum_k beta(q,n,k)
    = -10 log10 P(X >= n) = n q exactly (the tail ratios telescope)"""
    for n, q in ((5, 20), (100, 30), (255, 13), (255, 63)):
        m = oracle_errmod(oracle, 0.0, [q << 5 | 0] * n)
        assert abs(m[1, 1] - n * q) <= 2e-4 * n * q, (n, q, m[1, 1])
        assert m[0, 0] == 0.0


def test_depth_above_255_is_a_random_subsample_of_255(oracle):
    """errmod_cal() shuffles a deeper pileup (ks_shuffle on htslib's private rand48 stream) and uses its first 255 reads:
    the result must be the model's value for SOME 255-subset, here checked through its base counts: with two bases the
    heterozygote term gives c_A + c_C = 255 away, and repeated calls from the same stream state agree"""
    n = 400
    codes = [20 << 5 | (0 if i % 2 else 1) for i in range(n)]          # 200 A, 200 C
    a, b = oracle_errmod(oracle, 0.17, codes), oracle_errmod(oracle, 0.17, codes)
    assert np.array_equal(a, b)
    full = [model_errmod(0.17, [20 << 5 | 0] * k + [20 << 5 | 1] * (255 - k))[0, 1] for k in range(90, 166)]
    assert min(abs(a[0, 1] - f) for f in full) < 1e-3 * a[0, 1]        # equals the value of one split k : 255 - k
