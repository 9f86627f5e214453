"""INFO/I16 fields 13-16 in VGL_RNG_TILE (include/vcfgl_hip.h, "I16 tail distances"; the reference: vcfgl.cpp:647-663, 1029-1071, rng.h:12).

The specification is restated here in Python from the header's text alone -- second rand48 sequence from X0 = 0x7A11D157A11D, evaluation e = H(site) N +
sample owns draws [e block, (e + 1) block), read r takes draw r as a 31-bit integer x (the state's top 31 bits), tail = min(1 + x / (RAND_MAX / 50 + 1), 25),
credited to the base of the site's last simulated read, float32 sums in (sample, read) order -- and compared with the oracle's I16 field by field; the device
is compared with the oracle in tests/test_gpu_parity.py.  Also: the serial mode's tail distances still follow libc rand() (the golden VCFs pin those), and
the tile-mode values do not depend on the tiling."""
import ctypes as C

import numpy as np
import pytest

import synth
from test_rng_windows_cpu import site_hash
from vcfgl_amd import _abi
from vcfgl_amd.params import VcfglArgs

M48 = (1 << 48) - 1
A_, C_ = 0x5DEECE66D, 0xB
TAIL_X0 = 0x7A11D157A11D


def jump(x, n):
    a, c, ra, rc = A_, C_, 1, 0
    while n:
        if n & 1:
            ra, rc = (ra * a) & M48, (rc * a + c) & M48
        c = ((a + 1) * c) & M48
        a = (a * a) & M48
        n >>= 1
    return (ra * x + rc) & M48


@pytest.mark.parametrize("kw,N", [(dict(depth=7.0), 37), (dict(depth=20.0, error_qs=2, beta_variance=1e-5), 70), (dict(depth=3.0, gl_model=1), 5)])
def test_tile_mode_tail_distances_follow_the_header(oracle, kw, N):
    args = VcfglArgs(seed=9, error_rate=0.02, add_i16=1, add_qs=1, **kw)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    lib = _abi.load_library()
    p, _keep = args.to_struct(N)
    lay = _abi.RngLayout()
    assert lib.vgl_default_rng_layout(C.byref(p), C.byref(lay)) == 0
    mx = C.c_int64()
    assert lib.vgl_rng_tile_max_sites(C.byref(p), C.byref(mx)) == 0
    W, block = mx.value.bit_length() - 1, lay.block
    S, site0, cap = 12, 1000, 64
    gt = synth.acgt_sites(S, N, seed=N, missing=0.1)
    gt[5] = 0xFF                                                        # no reads at all
    o = oracle.Oracle(args, N)
    t = o.simulate(site0, gt, fields=["fmt_dp", "i16", "info_dp"], read_capacity=cap)
    dp, reads, i16, a2b = t.numpy("fmt_dp"), t.numpy("reads"), t.numpy("i16"), t.numpy("alleles2acgt")
    assert int(dp.max()) <= cap
    seen = 0
    for ls in range(S):
        if not i16[ls].any():                                           # (I16 is written for sites with reads and more than one allele)
            continue
        seen += 1
        h = site_hash(site0 + ls, W)
        s1 = s2 = np.float32(0)
        last = -1
        for s in range(N):
            st = jump(TAIL_X0, (h * N + s) * block)
            for r in range(int(dp[ls, s])):
                st = (st * A_ + C_) & M48
                td = min(1 + (st >> 17) // (2147483647 // 50 + 1), 25)
                s1 = np.float32(s1 + np.float32(td)); s2 = np.float32(s2 + np.float32(td * td))
            if dp[ls, s] > 0:
                last = int(reads[dp[ls, s] - 1, ls, s]) & 3
        assert last >= 0
        nA = int(t.numpy("n_alleles")[ls]); nObs = int(t.numpy("n_alleles_obs")[ls])
        want = [0.0, 0.0, 0.0, 0.0]
        if int(a2b[ls, 0]) == last:
            want[0], want[1] = float(s1), float(s2)
        for a in range(1, nA):
            if a == nObs:
                continue
            if int(a2b[ls, a]) == last:
                want[2] += float(s1); want[3] += float(s2)
        assert [float(x) for x in i16[ls, 12:]] == want, (ls, i16[ls, 12:], want, last)
    assert seen >= 6 and not i16[5].any()
    # independent of the tiling
    o2 = oracle.Oracle(args, N)
    parts = [o2.simulate(site0 + k, gt[k:k + 5], fields=["i16"]).numpy("i16") for k in range(0, S, 5)]
    assert np.array_equal(np.concatenate(parts).view(np.uint32), i16.view(np.uint32))


def test_serial_mode_still_draws_from_libc_rand(oracle):
    """VGL_RNG_SERIAL: the never-seeded libc rand() of the reference (the golden VCFs with -addI16 pin those values); tile mode differs from it"""
    args = VcfglArgs(seed=9, depth=6.0, error_rate=0.02, add_i16=1)
    gt = synth.acgt_sites(6, 20, seed=2)
    out = {}
    for mode in (_abi.VGL_RNG_SERIAL, _abi.VGL_RNG_TILE):
        a = VcfglArgs(**{**args.__dict__})
        a.rng_mode, a.beta_sampler = mode, _abi.VGL_BETA_RAND48
        o = oracle.Oracle(a, 20)
        out[mode] = (o.simulate(0, gt, fields=["i16"]).numpy("i16"), o.n_draw_rand() if hasattr(o, "n_draw_rand") else None)
    ser, til = out[_abi.VGL_RNG_SERIAL][0], out[_abi.VGL_RNG_TILE][0]
    assert ser[:, 12:].any() and til[:, 12:].any() and not np.array_equal(ser[:, 12:], til[:, 12:])
