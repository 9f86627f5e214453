"""VGL_RNG_SERIAL with --error-qs 2 and the default beta sampler: the device resolves the single std::mt19937
stream of the reference (rng.h:353-421) as a chain over the generator's output (vgl_betachain.hip).  Every deviate
of a tile is compared with a CPU emulation of std::mt19937 + libstdc++ gamma_distribution on fresh objects, for
shape parameters below and above 1, one chunk and many small chunks, one tile and consecutive tiles."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import synth
from vcfgl_amd import Simulator, VcfglArgs, _abi

pytestmark = pytest.mark.gpu


class StdBeta:
    """std::mt19937(seed) -> std::generate_canonical<double,53> -> std::gamma_distribution (Marsaglia-Tsang on
    polar normal pairs, the second deviate of a pair is kept inside one gamma_distribution object)"""

    def __init__(self, seed, n_words):
        mt = np.random.MT19937()
        mt._legacy_seeding(seed & 0xFFFFFFFF)
        self.words = mt.random_raw(n_words).astype(np.uint64)
        self.pos = 0

    def canonical(self):
        lo, hi = float(self.words[self.pos]), float(self.words[self.pos + 1])
        self.pos += 2
        r = (lo + hi * 4294967296.0) / 18446744073709551616.0
        return r if r < 1.0 else math.nextafter(1.0, 0.0)

    def gamma(self, alpha):
        malpha = alpha + 1.0 if alpha < 1.0 else alpha
        a1 = malpha - 1.0 / 3.0
        a2 = 1.0 / math.sqrt(9.0 * a1)
        saved = None
        while True:
            while True:
                if saved is not None:
                    n, saved = saved, None
                else:
                    while True:
                        x = 2.0 * self.canonical() - 1.0
                        y = 2.0 * self.canonical() - 1.0
                        r2 = x * x + y * y
                        if not (r2 > 1.0 or r2 == 0.0):
                            break
                    mult = math.sqrt(-2 * math.log(r2) / r2)
                    saved, n = x * mult, y * mult
                v = 1.0 + a2 * n
                if v > 0.0:
                    break
            v = v * v * v
            u = self.canonical()
            if not (u > 1.0 - 0.0331 * n * n * n * n and math.log(u) > 0.5 * n * n + a1 * (1.0 - v + math.log(v))):
                break
        if alpha == malpha:
            return a1 * v
        u = self.canonical()
        while u == 0.0:
            u = self.canonical()
        return u ** (1.0 / alpha) * a1 * v

    def beta(self, a, b):
        x = self.gamma(a)
        y = self.gamma(b)
        return x / (x + y)


def device_deviates(sim, n_max):
    buf = (C.c_double * n_max)()
    sim.lib.vgl_dbg_chain.restype = C.c_longlong
    sim.lib.vgl_dbg_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong]
    r = sim.lib.vgl_dbg_chain(sim.ctx, buf, n_max)
    assert 0 <= r <= n_max
    return np.array(buf[:r])


@pytest.mark.parametrize("err,var,max_words", [(0.01, 0.00198, None), (0.01, 1e-5, None), (0.2, 0.032, 40000), (0.01, 1e-5, 70000)])
def test_every_deviate_matches_the_cpu_generator(err, var, max_words, monkeypatch):
    if max_words:
        monkeypatch.setenv("VGL_CHAIN_MAX_WORDS", str(max_words))      # several chunks, the chain carried from one to the next
    else:
        monkeypatch.delenv("VGL_CHAIN_MAX_WORDS", raising=False)
    N, S, seed = 130, 24, 77
    args = VcfglArgs(seed=seed, depth=5.0, error_rate=err, error_qs=2, beta_variance=var)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_SERIAL, _abi.VGL_BETA_STD
    k = err * (1 - err) / var - 1.0
    a, b = err * k, (1 - err) * k
    sim = Simulator(args, N, max_sites_per_tile=S, hooks=True)      # vgl_dbg_chain / VGL_CHAIN_MAX_WORDS: the -DVGL_TEST_HOOKS build
    cpu = StdBeta(seed, 1_500_000)
    total = 0
    for tile in range(3):                                           # consecutive tiles continue the generator
        gt = synth.acgt_sites(S, N, seed=tile, missing=0.05)
        got = sim.simulate(tile * S, gt, fields=["fmt_dp"])
        dev = device_deviates(sim, 200000)
        assert len(dev) == int(got.numpy("fmt_dp").sum()) > 5000
        want = np.array([cpu.beta(a, b) for _ in range(len(dev))])
        rel = np.abs(want - dev) / np.maximum(np.abs(want), 1e-300)
        assert np.all(rel <= 1e-12), (tile, int(np.argmax(rel > 1e-12)), int(np.sum(rel > 1e-12)))
        total += len(dev)
    sim.close()
    assert total > 20000
