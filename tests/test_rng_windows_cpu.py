"""VGL_RNG_TILE window addressing (include/vcfgl_hip.h, vgl_rng_layout): evaluation (site, sample) owns the draws
[e block, (e + 1) block) of ONE rand48 sequence, e = H(site) n_samples + sample.

rand48 is X <- a X + c mod 2^48 with a^(2^k) = 1 mod 2^(k+2): states at offsets n, n + D, n + 2D with D a multiple of a high
power of two are linearly dependent, X(n) - 2 X(n + D) + X(n + 2D) = (a^D - 1)^2 X(n) + const.  Without the site permutation
H, sites s, s + 2^k, s + 2^(k+1) sit at such offsets (D = 2^k N block) and the second difference of the SAME (sample, stream,
draw index) collapses to 64 / 16 / 4 / 1 values at k = 16 .. 19 for C3's shape (round-2 review).  These tests pin (1) that H is
a permutation with H(0) = 0, identical in the library, the oracle and the restatement below; (2) that the lattice is there
without H (so the test would see a regression) and gone with it for k = 14 .. 22; (3) the oracle's tile-mode depths carry no
lag-2^k serial correlation."""
import ctypes as C
import random

import numpy as np
import pytest

import synth
from vcfgl_amd import _abi
from vcfgl_amd.params import VcfglArgs

M48 = (1 << 48) - 1


def site_hash(x, W):
    """restatement of the specification in include/vcfgl_hip.h"""
    if W <= 1:
        return x
    mask, sh = (1 << W) - 1, (W + 1) // 2
    x ^= x >> sh
    x = (x * 0xBF58476D1CE4E5B9) & mask
    x ^= x >> sh
    x = (x * 0x94D049BB133111EB) & mask
    x ^= x >> sh
    return x


def c3_args():
    a = VcfglArgs(seed=42, depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5)
    a.rng_mode, a.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    return a


def test_site_hash_is_a_permutation_fixing_zero():
    for W in range(0, 15):
        img = [site_hash(i, W) for i in range(1 << W)]
        assert sorted(img) == list(range(1 << W)), W
        assert img[0] == 0
    for W in (20, 24, 25, 30, 37, 40):                                   # large domains: injective on a sample, in range
        rnd = random.Random(W)
        xs = {rnd.randrange(1 << W) for _ in range(20000)}
        ys = {site_hash(x, W) for x in xs}
        assert len(ys) == len(xs) and max(ys) < (1 << W) and site_hash(0, W) == 0


def test_library_oracle_and_specification_agree(oracle):
    lib, ol = _abi.load_library(), oracle.lib()
    ol.vgl_oracle_site_hash.restype = C.c_uint64
    ol.vgl_oracle_site_hash.argtypes = [C.c_uint64, C.c_int]
    ol.vgl_oracle_site_hash_bits.argtypes = [C.c_uint64, C.c_uint64]
    rnd = random.Random(7)
    for a, N in ((c3_args(), 1000), (VcfglArgs(seed=1, depth=5.0, error_rate=0.01), 500), (VcfglArgs(seed=1, depth=300.0, error_rate=0.01), 3),
                 (VcfglArgs(seed=1, depth=30.0, error_rate=0.01, error_qs=2, beta_variance=1e-5), 2000)):
        p, _ = a.to_struct(N)
        lay = _abi.RngLayout()
        assert lib.vgl_default_rng_layout(C.byref(p), C.byref(lay)) == 0
        mx = C.c_int64()
        assert lib.vgl_rng_tile_max_sites(C.byref(p), C.byref(mx)) == 0
        W = mx.value.bit_length() - 1
        assert mx.value == 1 << W and mx.value * N * lay.block <= 1 << 48 < 2 * mx.value * N * lay.block
        assert ol.vgl_oracle_site_hash_bits(lay.block, N) == W
        h = C.c_int64()
        for site in [0, 1, 2, mx.value - 1] + [rnd.randrange(mx.value) for _ in range(2000)]:
            assert lib.vgl_rng_tile_site_hash(C.byref(p), site, C.byref(h)) == 0
            assert h.value == site_hash(site, W) == ol.vgl_oracle_site_hash(site, W)
        assert lib.vgl_rng_tile_site_hash(C.byref(p), mx.value, C.byref(h)) == _abi.VGL_E_ARG
        assert lib.vgl_rng_tile_site_hash(C.byref(p), -1, C.byref(h)) == _abi.VGL_E_ARG


@pytest.mark.parametrize("hashed", [False, True])
def test_second_difference_of_aligned_sites(oracle, hashed):
    """2 000 random (site, sample, stream, draw) per k: the 48-bit state of the same draw at sites s, s + 2^k, s + 2^(k+1).
    Unhashed spacing shows the lattice (<= 64 distinct second differences from k = 16 on); the library's addressing must give
    >= 1 900 distinct values for every k = 14 .. 22."""
    ol = oracle.lib()
    jump = ol.vgl_oracle_rand48_jump
    x0 = ol.vgl_oracle_rand48_seed(42)
    lib = _abi.load_library()
    p, _ = c3_args().to_struct(1000)
    lay = _abi.RngLayout()
    lib.vgl_default_rng_layout(C.byref(p), C.byref(lay))
    mx = C.c_int64()
    lib.vgl_rng_tile_max_sites(C.byref(p), C.byref(mx))
    W, N, block = mx.value.bit_length() - 1, 1000, lay.block
    offs = [lay.off[k] for k in range(4)]

    def state(site, smp, off):
        sp = site_hash(site, W) if hashed else site
        return jump(x0, (sp * N + smp) * block + off)

    rnd = random.Random(3)
    for k in range(14, 23):
        vals = set()
        for _ in range(2000):
            s = rnd.randrange(0, (1 << W) - (2 << k))
            smp = rnd.randrange(N)
            off = offs[rnd.randrange(4)] + 1 + rnd.randrange(40)         # first draws of the depth / haplotype / base / quality-score streams
            a, b, c = state(s, smp, off), state(s + (1 << k), smp, off), state(s + (2 << k), smp, off)
            vals.add((a - 2 * b + c) & M48)
        if hashed:
            assert len(vals) >= 1900, (k, len(vals))
        elif k >= 16:
            assert len(vals) <= 64, (k, len(vals))                       # the artefact this addressing removes


def test_first_draws_of_site_zero_are_the_reference_streams(oracle):
    """H(0) = 0: site 0 / sample 0 starts on the first draws of the seeded generator (test18: --depth 100 --seed 42 -> DP 85)"""
    a = VcfglArgs(seed=42, depth=100.0, error_rate=0.01)
    a.rng_mode = _abi.VGL_RNG_TILE
    t = oracle.Oracle(a, 1).simulate(0, np.zeros((1, 1), dtype=np.uint8), fields=["fmt_dp"])
    assert int(t.numpy("fmt_dp")[0, 0]) == 85


def test_oracle_depths_have_no_lag_2k_serial_correlation(oracle):
    """fmt_dp of sample j at sites s and s + 2^k, k = 14 .. 22 (C3's shape, 200 samples of the 1000 x 256 sites per lag):
    correlation within 5 sigma of 0, and the pair (dp(s), dp(s + 2^k)) is not a function of one another"""
    a = c3_args()
    N, S = 1000, 64
    o = oracle.Oracle(a, N)
    gt = np.zeros((S, N), dtype=np.uint8)
    base = o.simulate(1 << 10, gt, fields=["fmt_dp"]).numpy("fmt_dp").astype(np.float64)
    for k in range(14, 23):
        other = o.simulate((1 << 10) + (1 << k), gt, fields=["fmt_dp"]).numpy("fmt_dp").astype(np.float64)
        x, y = base.ravel() - base.mean(), other.ravel() - other.mean()
        r = float((x * y).sum() / np.sqrt((x * x).sum() * (y * y).sum()))
        assert abs(r) < 5.0 / np.sqrt(x.size), (k, r)
        assert (base != other).mean() > 0.85
