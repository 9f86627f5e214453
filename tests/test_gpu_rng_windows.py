"""VGL_RNG_TILE window addressing on the device (k_sitebase + the kernels that start from its states): the generator state in
front of every site's windows equals the specification's jump J^(block N H(site)) (x0), the states of sites s, s + 2^k,
s + 2^(k+1) carry no lattice (tests/test_rng_windows_cpu.py explains it), and far-apart sites agree with the oracle."""
import ctypes as C
import random

import numpy as np
import pytest

import synth
from test_rng_windows_cpu import M48, c3_args, site_hash
from vcfgl_amd import Simulator, VcfglArgs, _abi

pytestmark = pytest.mark.gpu


def _site_bases(sim, site0, n):
    gt = np.zeros((n, sim.n_samples), dtype=np.uint8)
    sim.simulate(site0, gt, fields=["fmt_dp"])
    out = (C.c_uint64 * n)()
    sim.lib.vgl_dbg_site_base.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    assert sim.lib.vgl_dbg_site_base(sim.ctx, out, n) == 0
    return list(out)


def test_device_site_states_follow_the_specification_and_carry_no_lattice(oracle):
    ol = oracle.lib()
    jump, x0 = ol.vgl_oracle_rand48_jump, ol.vgl_oracle_rand48_seed(42)
    N = 1000
    sim = Simulator(c3_args(), N, device=0, max_sites_per_tile=4, hooks=True)
    mx = C.c_int64()
    sim.lib.vgl_rng_tile_max_sites(C.byref(sim.params), C.byref(mx))
    W, block = mx.value.bit_length() - 1, sim.params.layout.block
    if block == 0:
        lay = _abi.RngLayout()
        sim.lib.vgl_default_rng_layout(C.byref(sim.params), C.byref(lay))
        block = lay.block
    rnd = random.Random(11)
    hap_off = 64 + 1                                                     # off[1] of the default layout, first haplotype draw
    for k in range(14, 23):
        vals_base, vals_hap = set(), set()
        for t in range(2000):                                            # (three one-site tile launches per trial)
            s = rnd.randrange(0, (1 << W) - (2 << k))
            st = [_site_bases(sim, s + d, 1)[0] for d in (0, 1 << k, 2 << k)]
            for d, got in zip((0, 1 << k, 2 << k), st):
                assert got == jump(x0, site_hash(s + d, W) * N * block), (s, d)
            smp = rnd.randrange(N)
            hp = [jump(x, smp * block + hap_off) for x in st]            # first haplotype state of sample smp at the three sites
            vals_base.add((st[0] - 2 * st[1] + st[2]) & M48)
            vals_hap.add((hp[0] - 2 * hp[1] + hp[2]) & M48)
        assert len(vals_base) >= 1900 and len(vals_hap) >= 1900, (k, len(vals_base), len(vals_hap))
    sim.close()


@pytest.mark.parametrize("kw", [dict(depth=20.0, error_qs=2, beta_variance=1e-5), dict(depth=20.0), dict(depth=5.0, do_unobserved=2),
                                dict(depth=300.0, gl_model=1)])
def test_sites_2k_apart_equal_the_oracle_and_are_uncorrelated(oracle, kw):
    """every field at sites 2^k apart (k = 14 .. 22 and the last sites of the addressable range) equals the oracle; fmt_dp at
    lag 2^k is uncorrelated"""
    N, S = (8 if kw.get("gl_model") == 1 else 320), 24
    args = VcfglArgs(seed=42, error_rate=0.01, add_pl=1, add_fmt_ad=1, **kw)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    sim = Simulator(args, N, device=0, max_sites_per_tile=S)
    orc = oracle.Oracle(args, N)
    mx = C.c_int64()
    sim.lib.vgl_rng_tile_max_sites(C.byref(sim.params), C.byref(mx))
    fields = ["site_status", "n_alleles", "alleles2acgt", "fmt_dp", "fmt_ad", "pl", "gl"]
    base_site = 12345
    tiles = {}
    for site0 in [base_site] + [base_site + (1 << k) for k in range(14, 23) if base_site + (1 << k) + S <= mx.value] + [mx.value - S]:
        gt = synth.binary_sites(site0, S, N)
        got, want = sim.simulate(site0, gt, fields=fields), orc.simulate(site0, gt, fields=fields)
        for f in fields:
            a, b = got.numpy(f), want.numpy(f)
            if a.dtype == np.float32:
                a, b = a.view(np.uint32), b.view(np.uint32)
            assert np.array_equal(a, b), (site0, f)
        tiles[site0] = got.numpy("fmt_dp").astype(np.float64)
    x = tiles[base_site]
    for site0, y in tiles.items():
        if site0 == base_site or kw["depth"] > 100:
            continue
        xc, yc = x.ravel() - x.mean(), y.ravel() - y.mean()
        r = float((xc * yc).sum() / np.sqrt((xc * xc).sum() * (yc * yc).sum()))
        assert abs(r) < 5.0 / np.sqrt(xc.size), (site0, r)
    sim.close()
