"""Parity of the HIP path (through the C ABI) with the CPU oracle on identical seeded inputs,
VGL_RNG_TILE addressing.  Integer fields must be bit-exact.  GL is bit-exact wherever the
per-read terms come from constants or the qScore LUT.  Where the device evaluates a logarithm or a power itself
(--precise-gl 1: log10_unit, three per read; GP: exp10_nonpos -- both within ~1e-15 relative of glibc's) a float32 result may
fall on the other side of a rounding boundary: GL then has to be within ONE unit in the last place of float32 of the oracle's,
with at most 1e-5 of the values not identical (expected ~1e-7: the count is printed), and GP within 1e-6 absolute
(`north_star`'s tolerance).  tests/test_gpu_scale_oracle.py applies the same criterion to 1.5e8 GL values."""
import numpy as np
import pytest

import golden_util as gu
import synth
from vcfgl_amd import Simulator, VcfglArgs, _abi

pytestmark = pytest.mark.gpu

INT_FIELDS = ["site_status", "n_alleles", "n_alleles_obs", "alleles2acgt", "info_dp", "info_ad", "info_adf",
              "info_adr", "fmt_dp", "pl", "fmt_ad", "fmt_adf", "fmt_adr"]
TOL = 1e-6


def ulps32(a, b):
    """distance of two finite float32 arrays in units in the last place (monotone integer image of the floats)"""
    def key(x):
        i = x.view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)
    return np.abs(key(np.ascontiguousarray(a)) - key(np.ascontiguousarray(b)))


def bits(a):
    """missing floats are NaN patterns (bcf_float_missing): compare raw bits"""
    return a.view(np.uint32) if a.dtype == np.float32 else a


def run_both(oracle, args, gt, site0=0, read_capacity=0, max_sites=None, hooks=False):
    """hooks=True: through the -DVGL_TEST_HOOKS build of the library (the VGL_DEBUG_* / VGL_NO_* environment overrides only exist there)"""
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    n_sites, N = gt.shape
    o = oracle.Oracle(args, N)
    sim = Simulator(args, N, device=0, max_sites_per_tile=max_sites or max(n_sites, 1), hooks=hooks)
    want = o.simulate(site0, gt, fields=sim.default_fields(), read_capacity=read_capacity)
    got = sim.simulate(site0, gt, read_capacity=read_capacity)
    sim.close()
    return want, got


def assert_parity(want, got, exact_gl=True, i16=False, qs=True, check_gp=True):
    for f in INT_FIELDS:
        assert np.array_equal(want.numpy(f), got.numpy(f)), f
    wgl, ggl = want.numpy("gl"), got.numpy("gl")
    wb, gb = wgl.view(np.uint32), ggl.view(np.uint32)
    miss = wb == _abi.FLOAT_MISSING_BITS
    assert np.array_equal(miss, gb == _abi.FLOAT_MISSING_BITS), "GL missing pattern"
    if exact_gl:
        assert np.array_equal(wb, gb), f"GL not bit-exact: {np.sum(wb != gb)} of {wb.size} differ"
    else:
        a, b = wgl[~miss], ggl[~miss]
        fin = np.isfinite(a)
        assert np.array_equal(fin, np.isfinite(b)) and np.array_equal(a[~fin], b[~fin])
        d = ulps32(a[fin], b[fin])
        nonid = int((d > 0).sum())
        print(f"GL (device log10): {nonid} of {d.size} values not identical to the oracle's, worst {int(d.max()) if d.size else 0} ulp")
        assert d.size == 0 or d.max() <= 1, f"GL {int(d.max())} units in the last place from the oracle"
        assert nonid <= max(1, int(np.ceil(1e-5 * d.size))), f"{nonid} of {d.size} GL values differ from the oracle"
    if check_gp:
        wgp, ggp = want.numpy("gp"), got.numpy("gp")
        m = wgp.view(np.uint32) == _abi.FLOAT_MISSING_BITS
        assert np.array_equal(m, ggp.view(np.uint32) == _abi.FLOAT_MISSING_BITS)
        dg = np.abs(wgp[~m].astype(np.float64) - ggp[~m].astype(np.float64))
        nonid = int((wgp[~m].view(np.uint32) != ggp[~m].view(np.uint32)).sum())
        if nonid:
            print(f"GP (device 10^x): {nonid} of {dg.size} values not identical to the oracle's, worst {dg.max():.3g} absolute")
        assert np.all(dg <= TOL)
        assert nonid <= max(1, int(np.ceil(1e-5 * dg.size))), f"{nonid} of {dg.size} GP values differ from the oracle"
    if qs and "qs" in got.arrays:
        assert np.array_equal(want.numpy("qs").view(np.uint32), got.numpy("qs").view(np.uint32)), "QS"
    if i16 and "i16" in got.arrays:
        assert np.array_equal(want.numpy("i16").view(np.uint32), got.numpy("i16").view(np.uint32)), "I16"      # all sixteen fields, both RNG modes (round 6: k_tail)


ALLTAGS = dict(add_gp=1, add_pl=1, add_qs=1, add_info_dp=1, add_fmt_ad=1, add_info_ad=1)
STRAND = dict(add_i16=1, add_fmt_adf=1, add_info_adf=1, add_fmt_adr=1, add_info_adr=1)


@pytest.mark.parametrize("N,n_sites", [(1, 5), (2, 7), (63, 9), (64, 9), (65, 9), (100, 40), (257, 11), (1000, 6)])
def test_gl2_fixed_q_shapes(oracle, N, n_sites):
    """ragged shapes: N below / at / above the 64-lane wavefront, single sample, many waves per site"""
    args = VcfglArgs(seed=42, depth=10, error_rate=0.01, **ALLTAGS)
    want, got = run_both(oracle, args, synth.binary_sites(3, n_sites, N), site0=3)
    assert_parity(want, got)


@pytest.mark.parametrize("depth", [0.0, 0.1, 2, 11.99, 12, 20, 30, 100])
def test_depth_branches(oracle, depth):
    """Poisson product branch (<12), rejection branch (>=12), empty sites (depth 0)"""
    args = VcfglArgs(seed=7, depth=depth, error_rate=0.002, **ALLTAGS)
    want, got = run_both(oracle, args, synth.binary_sites(0, 24, 130))
    assert_parity(want, got)
    if depth == 0.0:
        assert (got.numpy("site_status") == _abi.VGL_SITE_NO_READS).all()


@pytest.mark.parametrize("N,n_sites", [(1, 2500), (2, 1300), (3, 700), (7, 300), (63, 40), (100, 25), (1023, 5), (1024, 5), (1025, 5), (2047, 3), (2500, 3)])
def test_depth_kernel_chunks_span_sites(oracle, N, n_sites):
    """k_depth deals 1024 consecutive evaluations of the tile per wavefront: with few samples a chunk spans hundreds of sites,
    whose windows start at unrelated (hashed) positions -- the site of every dealt item is looked up (one compare for N >= 1024,
    a multiply-high below, N = 1 on its own)"""
    args = VcfglArgs(seed=3, depth=14.0, error_rate=0.01, add_fmt_ad=1)
    want, got = run_both(oracle, args, synth.binary_sites(77, n_sites, N), site0=77)
    for f in ("site_status", "fmt_dp", "info_dp", "fmt_ad"):
        assert np.array_equal(want.numpy(f), got.numpy(f)), f
    assert np.array_equal(want.numpy("gl").view(np.uint32), got.numpy("gl").view(np.uint32))


@pytest.mark.parametrize("chunk", [2048, 4096])
@pytest.mark.parametrize("N,n_sites", [(1, 9000), (3, 3000), (100, 90), (1000, 9), (2047, 5), (2048, 5), (2500, 4), (4095, 3), (4097, 3)])
def test_depth_kernel_large_chunks(oracle, monkeypatch, N, n_sites, chunk):
    """k_depth's chunk grows with the tile (2048 evaluations per wavefront from 2^23 evaluations, 4096 from 2^24): forced here on small
    tiles (hooks build) -- chunks spanning up to thousands of sites, sample counts around the chunk size (the one-compare site lookup
    needs N >= chunk, the multiply-high t < N + chunk), a ragged last chunk"""
    monkeypatch.setenv("VGL_DEPTH_CHUNK", str(chunk))
    args = VcfglArgs(seed=5, depth=13.0, error_rate=0.01, add_fmt_ad=1)
    want, got = run_both(oracle, args, synth.binary_sites(19, n_sites, N), site0=19, hooks=True)
    for f in ("site_status", "fmt_dp", "info_dp", "fmt_ad"):
        assert np.array_equal(want.numpy(f), got.numpy(f)), f
    assert np.array_equal(want.numpy("gl").view(np.uint32), got.numpy("gl").view(np.uint32))


@pytest.mark.parametrize("chunk", [1024, 4096])
def test_depth_kernel_per_sample_depths(oracle, monkeypatch, chunk):
    """k_depth<false>: every sample has its own mean depth of 12 or more (its own sq / alxm / g / e_hi, loaded when a lane takes an
    evaluation of that sample) -- the float64 exponent, both chunk sizes, chunks spanning many sites"""
    monkeypatch.setenv("VGL_DEPTH_CHUNK", str(chunk))
    N = 300
    rng = np.random.default_rng(chunk)
    depths = list(np.round(rng.uniform(12.0, 60.0, N), 3))
    depths[0], depths[1], depths[-1] = 12.0, 12.000001, 59.999
    args = VcfglArgs(seed=29, depths=depths, error_rate=0.01, add_fmt_ad=1)
    want, got = run_both(oracle, args, synth.binary_sites(7, 120, N), site0=7, hooks=True)
    for f in ("site_status", "fmt_dp", "info_dp", "fmt_ad"):
        assert np.array_equal(want.numpy(f), got.numpy(f)), f
    assert np.array_equal(want.numpy("gl").view(np.uint32), got.numpy("gl").view(np.uint32))


@pytest.mark.parametrize("depth", [12.0, 17.3, 48.0, 260.0])
def test_depth_kernel_without_the_exponent_table(oracle, monkeypatch, depth):
    """k_depth<ZT>: with one mean depth the acceptance bound's exponent comes from a float32 table (VglDevParams::pois_zt); with per-sample
    depths, and here with VGL_NO_POIS_ZT (hooks build), from the float64 expression -- the same draws either way"""
    args = VcfglArgs(seed=23, depth=depth, error_rate=0.01, add_fmt_ad=1)
    gt = synth.binary_sites(0, 40, 257)
    want, got = run_both(oracle, args, gt)
    monkeypatch.setenv("VGL_NO_POIS_ZT", "1")
    _, plain = run_both(oracle, args, gt, hooks=True)
    for f in ("fmt_dp", "info_dp", "fmt_ad"):
        assert np.array_equal(want.numpy(f), got.numpy(f)), f
        assert np.array_equal(plain.numpy(f), got.numpy(f)), f
    assert np.array_equal(want.numpy("gl").view(np.uint32), got.numpy("gl").view(np.uint32))


@pytest.mark.parametrize("du", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("rm", [(0, 0), (4, 1)])
def test_unobserved_and_skip_modes(oracle, du, rm):
    args = VcfglArgs(seed=11, depth=1.2, error_rate=0.05, do_unobserved=du, rm_invar_sites=rm[0], rm_empty_sites=rm[1], **ALLTAGS)
    want, got = run_both(oracle, args, synth.acgt_sites(60, 5, seed=du, missing=0.1))
    assert_parity(want, got)


@pytest.mark.parametrize("du", [0, 1, 3, 5])
@pytest.mark.parametrize("e,depth", [(0.0, 9), (0.03, 14), (0.35, 25)])
@pytest.mark.parametrize("eqs,precise", [(0, 0), (2, 0), (2, 1)])
def test_gl2_accumulator_classes(oracle, du, e, depth, eqs, precise):
    """GL model 2 keeps one accumulator per distinct update sequence (k_gl: 3 / 6 / 10 / 15 for 1 .. 4 distinct bases among an
    evaluation's reads, the loop chosen per wavefront, lanes ordered by that count): error rates from 0 (one or two bases per
    evaluation) to 0.35 (mostly four), with (du 1, 5) and without (du 0, 3) an allele that no read of the site shows, 4- and
    5-allele kernels, fixed / per-read / exact-log terms.  300 samples: several wavefronts per site and a ragged last one."""
    kw = dict(error_qs=2, beta_variance=1e-4) if eqs == 2 else {}
    if e == 0.0 and eqs == 2:
        pytest.skip("the beta sampler needs a mean error rate > 0")
    args = VcfglArgs(seed=31 + du, depth=depth, error_rate=e, do_unobserved=du, precise_gl=precise, add_pl=1, add_fmt_ad=1, **kw)
    want, got = run_both(oracle, args, synth.acgt_sites(12, 300, seed=7 + du, missing=0.02))
    assert_parity(want, got, exact_gl=(precise == 0), check_gp=False)


@pytest.mark.parametrize("e", [0.0, 0.2, 0.9])
def test_error_rates_incl_low_qscore(oracle, e):
    """e=0 -> q 63; e=0.9 -> q 0 (homT = -inf in the LUT)"""
    args = VcfglArgs(seed=5, depth=6, error_rate=e, **ALLTAGS)
    want, got = run_both(oracle, args, synth.acgt_sites(40, 70, seed=3))
    assert_parity(want, got, check_gp=(e != 0.9))


def test_strand_i16_adf_adr(oracle):
    args = VcfglArgs(seed=42, depth=5, error_rate=0.02, adjust_qs=3, **ALLTAGS, **STRAND)
    want, got = run_both(oracle, args, synth.acgt_sites(50, 66, seed=9, missing=0.05))
    assert_parity(want, got, i16=True)


@pytest.mark.parametrize("kw", [dict(depth=20.0, error_rate=0.01), dict(depth=3.0, error_rate=0.0), dict(depth=60.0, error_rate=0.9), dict(depth=150.0, error_rate=0.2),
                                dict(depth=8.0, error_rate=0.01, do_unobserved=0), dict(depth=8.0, error_rate=0.01, do_unobserved=3), dict(depth=20.0, error_rate=0.01, precise_gl=1),
                                dict(depth=20.0, error_rate=0.01, error_qs=1, beta_variance=1e-5), dict(depth=6.0, error_rate=0.002, adjust_qs=1)])
def test_gl2_one_base_run_table(oracle, kw, monkeypatch):
    """GL model 2 with one fixed score: a wavefront whose evaluations all show ONE base takes their accumulators from a table indexed by
    the depth (VglDevParams::gl2_run, built on the host by the reference's own update / maximum / subtract steps) instead of running
    the read loop.  Equal to the oracle, and to the library with the table switched off (VGL_NO_GL2_RUN, hooks build), for hom-ref
    tiles (every wavefront takes the table), mixed tiles, score 63 (e = 0), score 0 (e = 0.9: homT = -inf), depth 150, 4 / 5 alleles,
    sites whose only allele is the evaluation's (no absent allele: variant 1), --precise-gl 1 constants, --error-qs 1"""
    N = 200
    args = VcfglArgs(seed=17, **kw, **ALLTAGS)
    gt = synth.binary_sites(0, 24, N)
    gt[::2] = 0                                                      # hom-ref sites
    gt[1] = 0x11                                                     # hom-alt
    want, got = run_both(oracle, args, gt)
    assert_parity(want, got, exact_gl=not kw.get("precise_gl"), check_gp=(kw["error_rate"] != 0.9))
    monkeypatch.setenv("VGL_NO_GL2_RUN", "1")
    _, plain = run_both(oracle, args, gt, hooks=True)
    for f in ("gl", "pl", "fmt_ad"):
        assert np.array_equal(plain.numpy(f).view(np.uint32), got.numpy(f).view(np.uint32)), f


@pytest.mark.parametrize("ovc", [0, 1, 40])
@pytest.mark.parametrize("kw,N,n_sites", [(dict(depth=20.0, error_rate=0.01), 1000, 9), (dict(depth=20.0, error_rate=0.01), 1, 70), (dict(depth=14.0, error_rate=0.02), 63, 37),
                                          (dict(depth=30.0, error_rate=0.01, error_qs=2, beta_variance=1e-5), 1027, 5), (dict(depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5), 2500, 3),
                                          (dict(depth=60.0, error_rate=0.25), 300, 11), (dict(depth=150.0, error_rate=0.2, do_unobserved=0), 130, 6),
                                          (dict(depth=26.0, error_rate=0.02, error_qs=1, beta_variance=1e-4), 257, 12), (dict(depth=12.0, error_rate=0.0), 640, 4)])
def test_gl2_two_evaluations_per_thread(oracle, kw, N, n_sites, ovc, monkeypatch):
    """k_gl2 (GL model 2, three-kernel path, planes layout): a workgroup of 512 threads takes 1024 evaluations, the accumulators lie in
    a compact array (six full rows + nine rows of 256 columns (VGL_GL2_OVC) for the three- / four-base evaluations at the head of the sorted order),
    and a workgroup with more such evaluations than the upper rows hold is worked on again by k_gl's own body (k_gl2_scan, k_gl_redo).
    Forced on (VGL_GL2X=1, hooks build) for one fixed score and per-read scores, N = 1 ... 2500 (workgroups spanning sites, ragged
    ends), error rates that make most evaluations show three or four bases, sites without an absent allele; with the pool's limit at
    its default, at 1 (nearly every workgroup goes through k_gl_redo) and at 40.  Equal to the oracle, every tag."""
    args = VcfglArgs(seed=31, **kw, **ALLTAGS, **STRAND)
    gt = synth.acgt_sites(n_sites, N, seed=N + n_sites, missing=0.03)
    monkeypatch.setenv("VGL_GL2X", "2")                             # (2: also with GP and FORMAT/AD*, which the shipped choice leaves to k_gl)
    if ovc:
        monkeypatch.setenv("VGL_DEBUG_GL2_OVC", str(ovc))
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    sim = Simulator(args, N, device=0, max_sites_per_tile=n_sites, hooks=True)
    assert sim.info()["gl_wpb"] == 16, "k_gl2 is not what runs"
    sim.close()
    want, got = run_both(oracle, args, gt, site0=5, hooks=True)
    assert_parity(want, got, i16=True)


@pytest.mark.parametrize("N,n_sites", [(1, 40), (63, 17), (64, 16), (65, 15), (130, 1), (130, 33), (1000, 18), (1027, 5)])
@pytest.mark.parametrize("eqs", [0, 2])
def test_siteagg_sixteen_sites_per_wavefront(oracle, N, n_sites, eqs):
    """k_siteagg (INFO/QS, INFO/I16): a wavefront takes sixteen sites, walks the samples in chunks of 64 and keeps every order-dependent
    float32 sum in sample order -- ragged site counts, sample counts around the chunk size, skipped and empty sites, missing calls"""
    kw = dict(error_qs=2, beta_variance=1e-4, adjust_qs=2) if eqs else {}
    args = VcfglArgs(seed=9, depth=3.0, error_rate=0.02, rm_invar_sites=4, **kw, **ALLTAGS, **STRAND)
    want, got = run_both(oracle, args, synth.acgt_sites(n_sites, N, seed=N + n_sites, missing=0.04), site0=11)
    assert_parity(want, got, i16=True)


@pytest.mark.parametrize("kw,N,depth", [(dict(), 1, 4.0), (dict(), 63, 20.0), (dict(gl_model=1), 130, 9.0), (dict(error_qs=2, beta_variance=1e-5), 200, 20.0),
                                        (dict(error_qs=2, beta_variance=1e-5, precise_gl=1), 65, 14.0), (dict(error_qs=1, beta_variance=1e-5), 257, 3.0),
                                        (dict(gl_model=1, error_qs=2, beta_variance=1e-5, add_qs=1), 100, 30.0), (dict(depth=300.0, gl_model=1), 40, 300.0)])
def test_i16_tail_distances_in_tile_mode(oracle, kw, N, depth):
    """INFO/I16 fields 13-16 (vcfgl.cpp:647-663, 1029-1071) in VGL_RNG_TILE: one draw per read from the second rand48 sequence (k_tail, k_tail_fin) -- every
    sampler build that can run with -addI16 (fixed score with GL model 1: reads staged only for this; per-read scores; --precise-gl 1; a capacity beyond
    255 reads), all sixteen fields equal to the oracle, not zero, and independent of the tiling and of site0"""
    kw = dict(kw); kw.pop("depth", None)
    args = VcfglArgs(seed=11, depth=depth, error_rate=0.01, **kw, **STRAND)
    gt = synth.acgt_sites(24, N, seed=4, missing=0.1)
    gt[3] = 0xFF                                                # a site without reads: no draw, no base
    want, got = run_both(oracle, args, gt, site0=7)
    assert_parity(want, got, i16=True, qs=False, check_gp=False, exact_gl=not args.precise_gl)
    tail = got.numpy("i16")[:, 12:].astype(np.float64)
    ok = tail.sum(axis=1) > 0                                   # (I16 is written for sites with reads and more than one allele)
    assert ok.sum() >= 12 and (tail[3] == 0).all()
    # every draw of a site goes to ONE base (the last read's): fields 13 + 15 hold the site's sum; mean tail distance (1 + ... + 24 + 26 x 25) / 50 = 19
    nread = got.numpy("info_dp")[ok].sum()
    assert abs(tail[ok][:, [0, 2]].sum() / nread - 19.0) < 4 * 8.0 / np.sqrt(nread) + 0.05
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    sim = Simulator(args, N, device=0, max_sites_per_tile=5)
    parts = [sim.simulate(7 + s0, gt[s0:s0 + 5]).numpy("i16") for s0 in range(0, 24, 5)]
    sim.close()
    assert np.array_equal(np.concatenate(parts).view(np.uint32), got.numpy("i16").view(np.uint32))


@pytest.mark.parametrize("kw,N,depth", [(dict(error_rate=1e-7), 300, 20), (dict(error_rate=0.01, i16_mapq=37), 700, 20), (dict(error_rate=0.01, i16_mapq=59), 400, 60),
                                        (dict(error_rate=0.01, i16_mapq=0), 100, 5), (dict(error_rate=0.01, i16_mapq=60), 2000, 30)])
def test_siteagg_beyond_the_exact_range_of_float32(oracle, kw, N, depth):
    """INFO/I16 fields 5-12 are float32 running sums of integers; k_siteagg takes them from integer totals while every partial sum is
    exact (<= 2^24, or the multiples of 2^tz(c) up to 2^(24 + tz(c)) for the repeated addition of the mapping quality c) and walks the
    rest in the reference's order: q = 63 (sum of squares 3969 per read: 6000 reads cross 2^24), odd mapping qualities (37^2 = 1369:
    14000 reads; 59^2 x 24000), mapping quality 0, and a deep wide site (60^2 x 60000 = 2^27.7, still exact: 3600 = 2^4 x 225)"""
    args = VcfglArgs(seed=3, depth=depth, add_qs=1, **kw, **STRAND)
    gt = synth.binary_sites(0, 5, N)
    gt[1] = 0                                                   # a hom-ref site: every read on one base
    want, got = run_both(oracle, args, gt)
    assert float(want.numpy("i16")[:, 4:12].max()) > 2 ** 24 or kw.get("i16_mapq") == 0
    for f in ("site_status", "fmt_dp", "info_dp", "info_adf"):
        assert np.array_equal(want.numpy(f), got.numpy(f)), f
    assert np.array_equal(want.numpy("qs").view(np.uint32), got.numpy("qs").view(np.uint32)), "QS"
    assert np.array_equal(want.numpy("i16").view(np.uint32), got.numpy("i16").view(np.uint32)), "I16"


@pytest.mark.parametrize("precise", [0, 1])
@pytest.mark.parametrize("adj", [0, 3])
def test_error_qs2_beta_quality_scores(oracle, precise, adj):
    if precise and (adj & 1):
        pytest.skip("--adjust-qs 1 requires --precise-gl 0 (io.cpp)")
    args = VcfglArgs(seed=42, depth=8, error_rate=0.01, error_qs=2, beta_variance=1e-5, precise_gl=precise,
                     adjust_qs=adj, **ALLTAGS)
    want, got = run_both(oracle, args, synth.binary_sites(0, 30, 100), read_capacity=40)
    assert np.array_equal(want.numpy("reads"), got.numpy("reads")), "per-read base / qscore dump"
    assert_parity(want, got, exact_gl=not precise)


@pytest.mark.parametrize("mean,var", [(0.4, 0.1), (0.3, 0.15), (0.05, 0.03)])
def test_error_qs2_alpha_below_one(oracle, mean, var):
    """beta shapes with alpha < 1: the pow() branch of the gamma sampler (rng.h:146-148; doc/error_qs.MD documents
    beta(0.4, 0.1)).  The deviate itself carries ocml's double pow() instead of glibc's: the two differ by a few units
    in the last place in ~20 % of the draws (measured: <= 4.4e-16 relative, tools/alpha_lt1_diag.py), which can move
    (int)(-10 log10 p) only when p lies within that distance of a quality-score boundary -- about 1e-15 per read, none in
    the 3e5 reads measured.  Every integer field (per-read base and quality score included) must therefore be EQUAL here;
    the deviates are compared at 1e-15 relative."""
    args = VcfglArgs(seed=3, depth=4, error_rate=mean, error_qs=2, beta_variance=var, **ALLTAGS)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    gt = synth.binary_sites(0, 120, 64)
    want = oracle.Oracle(args, 64).simulate(0, gt, read_capacity=24, deviates=True)
    sim = Simulator(args, 64, device=0, max_sites_per_tile=120)
    got = sim.simulate(0, gt, read_capacity=24, deviates=True)
    sim.close()
    assert np.array_equal(want.numpy("reads"), got.numpy("reads")), "per-read base / qscore dump"
    have = want.numpy("reads") != 0xFF
    we, ge = want.numpy("read_errp")[have], got.numpy("read_errp")[have]
    assert np.all(np.abs(we - ge) <= 1e-15 * np.abs(we))
    assert_parity(want, got)


def test_error_qs1_site_level_beta(oracle):
    args = VcfglArgs(seed=42, depth=6, error_rate=0.05, error_qs=1, beta_variance=1e-3, **ALLTAGS)
    want, got = run_both(oracle, args, synth.binary_sites(0, 40, 90))
    assert_parity(want, got)


def test_qs_bins(oracle):
    bins = [(0, 2, 2), (3, 14, 12), (15, 30, 23), (31, 40, 37)]
    args = VcfglArgs(seed=42, depth=8, error_rate=0.01, error_qs=2, beta_variance=1e-5, qs_bins=bins, adjust_qs=3, **ALLTAGS)
    want, got = run_both(oracle, args, synth.binary_sites(0, 20, 100))
    assert_parity(want, got)


def test_per_sample_depths(oracle):
    N = 70
    depths = [0.1, 10, 25, 3] * 17 + [0.0, 40.0]
    args = VcfglArgs(seed=42, depths=depths, error_rate=0.01, **ALLTAGS)
    want, got = run_both(oracle, args, synth.binary_sites(0, 30, N))
    assert_parity(want, got)


@pytest.mark.parametrize("depth,du", [(1, 1), (10, 2), (30, 3)])
def test_gl_model1_fixed_q(oracle, depth, du):
    """config C2 family: -GL 1 (errmod) with one fixed qScore"""
    args = VcfglArgs(seed=42, depth=depth, error_rate=0.01, gl_model=1, do_unobserved=du, adjust_qs=1, **ALLTAGS)
    want, got = run_both(oracle, args, synth.binary_sites(0, 50, 100))
    assert_parity(want, got)


@pytest.mark.parametrize("depth,adj", [(2, 0), (12, 1), (40, 3)])
def test_gl_model1_per_read_q(oracle, depth, adj):
    """-GL 1 --error-qs 2: errmod over per-read quality scores (LDS histogram instead of a sort)"""
    args = VcfglArgs(seed=42, depth=depth, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=1, adjust_qs=adj, **ALLTAGS)
    want, got = run_both(oracle, args, synth.binary_sites(0, 30, 100))
    assert_parity(want, got)


def test_config_c4_flags(oracle):
    """BASELINE.json configs[3] flags at its sample count: 2000 samples, depth 30, --error-qs 2 with the
    rta3 quality-score bins (32 wavefronts per site)"""
    bins = [(0, 2, 2), (3, 14, 12), (15, 30, 23), (31, 40, 37)]
    args = VcfglArgs(seed=42, depth=30, error_rate=0.01, error_qs=2, beta_variance=1e-5, qs_bins=bins, **ALLTAGS)
    want, got = run_both(oracle, args, synth.binary_sites(123456, 6, 2000), site0=123456)
    assert_parity(want, got)


def test_config_c5_flags(oracle):
    """BASELINE.json configs[4] simulation flags: exploded (all hom-ref) sites, -doUnobserved 2, PL,
    depth 5, 500 samples; site indices near the end of a 50M-site job"""
    args = VcfglArgs(seed=42, depth=5, error_rate=0.01, do_unobserved=2, add_pl=1)
    gt = np.zeros((40, 500), np.uint8)
    want, got = run_both(oracle, args, gt, site0=49_999_000)
    assert_parity(want, got, check_gp=False, qs=False)
    assert (got.numpy("n_alleles") >= 2).all()


@pytest.mark.parametrize("seed", [-1, -123456789, 0, 2 ** 31 - 1])
def test_seed_range(oracle, seed):
    """io.cpp:1054-1061: the low 32 bits of the (signed) seed seed every stream"""
    args = VcfglArgs(seed=seed, depth=15, error_rate=0.02, error_qs=2, beta_variance=1e-4, **ALLTAGS)
    want, got = run_both(oracle, args, synth.acgt_sites(12, 80, seed=2, missing=0.02))
    assert_parity(want, got)


def test_qs_bin_miss_is_an_error(oracle):
    """apply_qs_bins() exits with "Could not find a range for qs value" (vcfgl.cpp:63): here VGL_E_QSBIN"""
    from vcfgl_amd import VglError
    args = VcfglArgs(seed=1, depth=10, error_rate=0.01, error_qs=2, beta_variance=1e-5, qs_bins=[(0, 10, 5)])
    sim = Simulator(args, 64, max_sites_per_tile=4)
    with pytest.raises(VglError) as ei:
        sim.simulate(0, synth.binary_sites(0, 4, 64))
    assert ei.value.code == _abi.VGL_E_QSBIN
    sim.close()
    with pytest.raises(oracle.OracleError) as eo:
        oracle.Oracle(args, 64).simulate(0, synth.binary_sites(0, 4, 64))
    assert eo.value.code == _abi.VGL_E_QSBIN


def test_bad_parameters_are_rejected_like_the_reference():
    from vcfgl_amd import VglError
    for kw, frag in ((dict(depth=5, error_rate=0.1, gl_model=1, precise_gl=1), b"not supported with genotype likelihood model 1"),
                     (dict(depth=5, error_rate=0.0, error_qs=2, beta_variance=1e-5), b"--error-rate"),
                     (dict(depth=5, error_rate=0.5, error_qs=1, beta_variance=0.5), b"shape parameters")):
        with pytest.raises(VglError) as ei:
            Simulator(VcfglArgs(seed=1, **kw), 4, max_sites_per_tile=2)
        assert ei.value.code == _abi.VGL_E_ARG and frag in str(ei.value).encode()


@pytest.mark.parametrize("eqs", [0, 2])
def test_maximum_depth(oracle, eqs):
    """--depth 500 is the reference's upper bound (shared.h:63, io.cpp:861): ~500 reads per sample,
    quality-score pools of > 6144 items per wavefront run in several LDS segments"""
    args = VcfglArgs(seed=42, depth=500, error_rate=0.01, error_qs=eqs, beta_variance=1e-5 if eqs else -1.0, add_pl=1, add_fmt_ad=1, add_qs=1)
    want, got = run_both(oracle, args, synth.binary_sites(0, 2, 70))
    assert_parity(want, got, check_gp=False)
    assert got.numpy("fmt_dp").max() > 500


def test_many_samples_and_empty_tile(oracle):
    """5000 samples (79 wavefronts per site); a zero-site tile is a no-op"""
    args = VcfglArgs(seed=42, depth=3, error_rate=0.01, add_pl=1)
    want, got = run_both(oracle, args, synth.binary_sites(7, 3, 5000), site0=7)
    assert_parity(want, got, check_gp=False, qs=False)
    sim = Simulator(args, 5000, max_sites_per_tile=4)
    t = sim.simulate(0, np.zeros((0, 5000), np.uint8))
    assert t.numpy("fmt_dp").shape == (0, 5000)
    sim.close()


def test_site_index_invariance(oracle):
    """tiles are addressed by absolute site index: splitting a run into tiles (or shards)
    does not change any value"""
    args = VcfglArgs(seed=42, depth=10, error_rate=0.01, **ALLTAGS)
    gt = synth.binary_sites(0, 64, 100)
    N = 100
    sim = Simulator(args, N, max_sites_per_tile=64)
    whole = sim.simulate(0, gt)
    a = sim.simulate(0, gt[:20])
    b = sim.simulate(20, gt[20:])
    sim.close()
    for f in ["fmt_dp", "gl", "pl", "fmt_ad", "info_ad", "alleles2acgt"]:
        assert np.array_equal(bits(whole.numpy(f)), bits(np.concatenate([a.numpy(f), b.numpy(f)]))), f


@pytest.mark.parametrize("name", ["test2", "test5", "test12", "test14", "test17", "test18"])
def test_reference_inputs_tile_mode(oracle, name):
    """the reference's own test inputs and flags (error-qs 0 cases), counter-addressed streams"""
    args, vcf, sites, gold = gu.load_case(name, rng_mode=_abi.VGL_RNG_TILE, beta_sampler=_abi.VGL_BETA_RAND48)
    gt = np.stack([s.gt for s in sites])
    want, got = run_both(oracle, args, gt)
    assert_parity(want, got, i16=bool(args.add_i16), qs=bool(args.add_qs), check_gp=bool(args.add_gp))


def test_first_evaluation_equals_reference_stream(oracle):
    """site 0 / sample 0 starts every stream at the reference's initial state, so its depth
    equals the reference's first depth draw: 85 for --depth 100 --seed 42 (test/reference/test18)."""
    args = VcfglArgs(seed=42, depth=100, error_rate=0.0, do_unobserved=5)
    sim = Simulator(args, 2, max_sites_per_tile=4)
    t = sim.simulate(0, np.zeros((1, 2), np.uint8))
    sim.close()
    assert t.numpy("fmt_dp")[0, 0] == 85


def test_device_buffers_and_stream(oracle):
    """device variant of the ABI: torch owns GT / outputs, launch on a side stream"""
    import torch
    args = VcfglArgs(seed=42, depth=10, error_rate=0.01, add_pl=1)
    args.rng_mode = _abi.VGL_RNG_TILE
    N, S = 200, 33
    gt = synth.binary_sites(0, S, N)
    sim = Simulator(args, N, max_sites_per_tile=S)
    tile = sim.new_tile(S, fields=["fmt_dp", "gl", "pl"], device="cuda:0")
    st = torch.cuda.Stream()
    dgt = torch.from_numpy(gt).to("cuda:0")
    torch.cuda.synchronize()
    sim.simulate_device(0, dgt, tile, stream=st.cuda_stream)
    sim.check(stream=st.cuda_stream)
    host = sim.simulate(0, gt, fields=["fmt_dp", "gl", "pl"])
    sim.close()
    for f in ["fmt_dp", "gl", "pl", "n_alleles"]:
        assert np.array_equal(bits(tile.numpy(f)), bits(host.numpy(f))), f


def test_capacity_sized_from_largest_mean(oracle):
    args = VcfglArgs(seed=42, depths=[0.0, 400.0], error_rate=0.01)
    sim = Simulator(args, 2, max_sites_per_tile=4)
    t = sim.simulate(0, np.zeros((2, 2), np.uint8))
    assert t.numpy("fmt_dp")[:, 1].min() > 300
    sim.close()


@pytest.mark.parametrize("kw,N,S", [(dict(), 100, 8), (dict(error_qs=2, beta_variance=1e-5), 100, 8), (dict(gl_model=1), 70, 5),
                                    (dict(error_qs=2, beta_variance=1e-5, precise_gl=1, add_pl=1), 65, 3), (dict(depths=1), 130, 4),
                                    (dict(error_qs=2, beta_variance=1e-5, add_qs=1, add_i16=1, add_fmt_adf=1, add_gp=1, add_fmt_ad=1), 90, 6),
                                    (dict(layout="sm"), 300, 2500)])
def test_a_draw_deeper_than_the_staging_capacity_runs_the_tile_again(oracle, kw, N, S, monkeypatch):
    """vcfgl grows its read buffers (bcf_utils.cpp:618-648); the library stages a fixed number of reads per sample (mean + 8 sigma + 16) and flags a tile
    with a deeper draw.  The host entry points then run that tile again on a sibling context with the staging layout's largest capacity, in sub-tiles
    -- VGL_DEBUG_READ_CAP=8 at depth 20 makes every tile such a tile: equal to the oracle, through vgl_simulate_tile and through the async pair
    (S = 2500: two sub-tiles of the sibling, sample-major slabs)"""
    import ctypes as C
    monkeypatch.setenv("VGL_DEBUG_READ_CAP", "8")
    kw = dict(kw)
    depths = [20.0 + (i % 7) for i in range(N)] if kw.pop("depths", 0) else None
    sm = kw.pop("layout", None) == "sm"
    args = VcfglArgs(seed=42, depth=20, depths=depths, error_rate=0.01, **kw)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    if sm:
        args.out_layout = _abi.VGL_LAYOUT_SAMPLE_MAJOR
    gt = synth.acgt_sites(S, N, seed=S, missing=0.03)
    sim = Simulator(args, N, max_sites_per_tile=S, hooks=True)
    assert sim.info()["read_cap"] == 8 and sim.info()["test_hooks"] == 1
    got = sim.simulate(3, gt)
    want = oracle.Oracle(args, N).simulate(3, gt, fields=sim.default_fields())
    assert int(want.numpy("fmt_dp").max()) > 8
    assert_parity(want, got, i16=True, exact_gl=not args.precise_gl)
    # the async pair: the rerun happens in vgl_tile_wait, the other slot's tile is untouched
    t0, t1 = sim.new_tile(S), sim.new_tile(S)
    k0, k1 = C.c_int32(), C.c_int32()
    sim._check(sim.lib.vgl_simulate_tile_async(sim.ctx, 3, S, gt.ctypes.data, t0.byref(), C.byref(k0)))
    sim._check(sim.lib.vgl_simulate_tile_async(sim.ctx, 3, S, gt.ctypes.data, t1.byref(), C.byref(k1)))
    sim._check(sim.lib.vgl_tile_wait(sim.ctx, k0)); sim._check(sim.lib.vgl_tile_wait(sim.ctx, k1))
    for t in (t0, t1):
        for f in ("fmt_dp", "gl", "site_status", "n_alleles"):
            assert np.array_equal(bits(t.numpy(f)), bits(got.numpy(f))), f
    sim.close()


@pytest.mark.parametrize("how", ["device", "serial", "dump"])
def test_capacity_overflow_is_reported_where_the_tile_cannot_be_run_again(oracle, how, monkeypatch):
    """... and where that cannot be done a depth beyond the staging capacity surfaces as VGL_E_CAPACITY, never as wrong data: the device-buffer entry
    point (the flags are read by vgl_ctx_check, which does not know the tile), VGL_RNG_SERIAL (the streams have moved on), a per-read dump (planes of the
    caller's own capacity)"""
    import torch
    from vcfgl_amd import VglError
    monkeypatch.setenv("VGL_DEBUG_READ_CAP", "8")
    args = VcfglArgs(seed=42, depth=20, error_rate=0.01)
    args.rng_mode = _abi.VGL_RNG_SERIAL if how == "serial" else _abi.VGL_RNG_TILE
    sim = Simulator(args, 100, max_sites_per_tile=8, hooks=True)
    gt = synth.binary_sites(0, 8, 100)
    with pytest.raises(VglError) as ei:
        if how == "device":
            tile = sim.new_tile(8, fields=["fmt_dp", "gl"], device="cuda:0")
            sim.simulate_device(0, torch.from_numpy(gt).to("cuda:0"), tile)
            sim.check()
        else:
            sim.simulate(0, gt, read_capacity=8 if how == "dump" else 0)
    assert ei.value.code == _abi.VGL_E_CAPACITY
    sim.close()


@pytest.mark.parametrize("serial", [0, 1])
def test_deviate_dumps_match_oracle(oracle, serial):
    """ABI 2: read_errp (error_qs 2) and site_pick_err (error_qs 1) are the doubles the reference prints with
    -printQsError / -printBasePickError; bit-equal to the oracle's in both RNG modes, with and without the
    --precise-gl staging planes."""
    gt = synth.binary_sites(0, 24, 70)
    for precise in (0, 1):
        args = VcfglArgs(seed=11, depth=5, error_rate=0.02, error_qs=2, beta_variance=1e-4, precise_gl=precise)
        args.rng_mode = _abi.VGL_RNG_SERIAL if serial else _abi.VGL_RNG_TILE
        args.beta_sampler = _abi.VGL_BETA_STD if serial else _abi.VGL_BETA_RAND48
        sim = Simulator(args, 70, device=0, max_sites_per_tile=24)
        want = oracle.Oracle(args, 70).simulate(0, gt, fields=sim.default_fields(), read_capacity=32, deviates=True)
        got = sim.simulate(0, gt, read_capacity=32, deviates=True)
        plain = sim.simulate(24, gt) if not serial else None          # a later tile without the dump still runs
        sim.close()
        dp = want.numpy("fmt_dp")
        assert np.array_equal(dp, got.numpy("fmt_dp")) and dp.max() <= 32
        live = np.arange(32)[:, None, None] < dp[None, :, :]
        w, g = want.numpy("read_errp"), got.numpy("read_errp")
        assert live.sum() > 1000
        if serial:      # libstdc++'s gamma_distribution: log / sqrt / pow of the device's libm, within 1e-12 of glibc's (as tests/test_gpu_betachain.py)
            assert np.all(np.abs(w[live] - g[live]) <= 1e-12 * np.abs(w[live]))
        else:
            assert np.array_equal(w[live].view(np.uint64), g[live].view(np.uint64))
        assert np.array_equal(want.numpy("reads"), got.numpy("reads"))
        assert plain is None or plain.numpy("fmt_dp").shape == dp.shape
    args = VcfglArgs(seed=11, depth=5, error_rate=0.05, error_qs=1, beta_variance=1e-3)
    args.rng_mode = _abi.VGL_RNG_SERIAL if serial else _abi.VGL_RNG_TILE
    args.beta_sampler = _abi.VGL_BETA_STD if serial else _abi.VGL_BETA_RAND48
    sim = Simulator(args, 70, device=0, max_sites_per_tile=24)
    want = oracle.Oracle(args, 70).simulate(0, gt, fields=sim.default_fields(), deviates=True)
    got = sim.simulate(0, gt, deviates=True)
    sim.close()
    reach = want.numpy("info_dp") > 0
    assert reach.sum() > 10
    w, g = want.numpy("site_pick_err")[reach], got.numpy("site_pick_err")[reach]
    assert np.all(np.abs(w - g) <= 1e-12 * np.abs(w)) if serial else np.array_equal(w.view(np.uint64), g.view(np.uint64))


@pytest.mark.parametrize("depth,N", [(8, 100), (70, 64)])
def test_undecided_quality_scores_are_redrawn_exactly(oracle, monkeypatch, depth, N):
    """k_sample<2> takes a read's quality score from the float32 error probability unless that sits within the
    float32 bound of an integer boundary (about 1 read in 10^4); such a read is drawn again in double by the lane
    that owns it.  VGL_DEBUG_QS_EXACT=1 sends EVERY read down that path (second case: several LDS segments per
    wavefront): results must not change."""
    monkeypatch.setenv("VGL_DEBUG_QS_EXACT", "1")
    args = VcfglArgs(seed=5, depth=depth, error_rate=0.01, error_qs=2, beta_variance=1e-5, adjust_qs=3, **ALLTAGS)
    want, got = run_both(oracle, args, synth.binary_sites(0, 12, N), read_capacity=128, hooks=True)
    assert np.array_equal(want.numpy("reads"), got.numpy("reads"))
    assert_parity(want, got)


@pytest.mark.parametrize("every,cap", [(0, None), (3, None), (7, None), (3, 16), (3, 256)])
@pytest.mark.parametrize("depth,N,bins", [(20, 300, False), (30, 130, True), (70, 64, False)])
def test_deferred_reads_are_redrawn_by_k_redo(oracle, monkeypatch, every, cap, depth, N, bins):
    """The default tag surface runs k_sample<2> without any double-precision fallback code: a read that one of the float32
    bounds cannot settle (pool loop: the two bounded log tests; dense pass: the quality score) is appended to a list and drawn
    again in double by k_redo, which patches the staged read.  VGL_DEBUG_REDO_EVERY=k sends every k-th candidate of each of the
    three sources down that path (0: only the genuine ones); the GLs -- which are all that depends on the scores here -- must not
    change.  Cases: one segment per wavefront, --qs-bins (k_redo applies them), several LDS segments per wavefront; cap 16: the
    list overflows and the rest of the reads travels through the bitmap (round 5: the list has 64 partitions with a counter each -- cap 16
    leaves them no entry at all, cap 256 four entries each, so that list and bitmap are both in use)."""
    if every:
        monkeypatch.setenv("VGL_DEBUG_REDO_EVERY", str(every))
    if cap is not None:
        monkeypatch.setenv("VGL_DEBUG_REDO_CAP", str(cap))
    kw = dict(qs_bins=[(0, 2, 2), (3, 14, 12), (15, 30, 23), (31, 40, 37)]) if bins else {}          # the rta3 bins (doc/error_qs.MD)
    args = VcfglArgs(seed=77, depth=depth, error_rate=0.01, error_qs=2, beta_variance=1e-5, add_pl=1, **kw)
    want, got = run_both(oracle, args, synth.binary_sites(0, 10, N), hooks=True)
    assert_parity(want, got, check_gp=False)


@pytest.mark.parametrize("period,period_n", [(3, 5), (1, 7), (5, 2), (8, 8)])
def test_bounded_tests_run_on_one_period(oracle, monkeypatch, period, period_n):
    """in the float32 pool loop a lane held for BOTH bounded logarithm tests advanced only when both counters fired -- every lcm of
    VGL_SLOW_PERIOD and VGL_SLOW_PERIOD_N under the hooks.  The loop now runs both tests on VGL_SLOW_PERIOD; whatever the two values, no candidate
    forced to k_redo (VGL_DEBUG_REDO_EVERY unset), the result is the oracle's (the A / B attempt bookkeeping -- useB, rejB, hold -- under periods
    other than the default 4 / 4)."""
    monkeypatch.setenv("VGL_SLOW_PERIOD", str(period))
    monkeypatch.setenv("VGL_SLOW_PERIOD_N", str(period_n))
    args = VcfglArgs(seed=19, depth=25, error_rate=0.01, error_qs=2, beta_variance=1e-5, add_pl=1)
    want, got = run_both(oracle, args, synth.binary_sites(0, 16, 200), hooks=True)
    assert_parity(want, got, check_gp=False)


@pytest.mark.parametrize("limit,pool", [(None, None), (1200, None), (1, None), (None, 640), (900, 1280), (None, 64)])
@pytest.mark.parametrize("depth,N,bins", [(20, 300, False), (30, 130, True), (20, 1000, False)])
def test_split_build_of_the_default_tag_surface(oracle, monkeypatch, limit, pool, depth, N, bins):
    """Round 6: k_sample<2, LEAN 2> runs as two kernels (k_sample_seg) -- <., 1> takes ONE pool segment per wavefront, a wavefront with more reads
    than the pool holds appends itself to a list and leaves; <., 2> runs the segment loop over the listed wavefronts.  Cases: the library's own
    limit (nothing listed), VGL_DEBUG_SEG_LIMIT 1200 (about a fifth of depth 20's wavefronts listed, each then served in one segment by the
    second kernel) and 1 (every wavefront listed), VGL_DEBUG_POOL_CAP 640 / 1280 / 64 with the split forced on (every listed wavefront really takes
    2 ... 20+ segments; 64: a segment per lane's worth of reads), 1000 samples (16 wavefronts per site, the last one ragged).  Every field equal to
    the oracle; with the hooks off the same cases run the shipped choice (test_deferred_reads_are_redrawn_by_k_redo)."""
    if limit is not None:
        monkeypatch.setenv("VGL_DEBUG_SEG_LIMIT", str(limit))
    if pool is not None:
        monkeypatch.setenv("VGL_DEBUG_POOL_CAP", str(pool))
        monkeypatch.setenv("VGL_SEG_SPLIT", "1")
    kw = dict(qs_bins=[(0, 2, 2), (3, 14, 12), (15, 30, 23), (31, 40, 37)]) if bins else {}
    args = VcfglArgs(seed=91, depth=depth, error_rate=0.01, error_qs=2, beta_variance=1e-5, add_pl=1, **kw)
    want, got = run_both(oracle, args, synth.binary_sites(0, 12, N), hooks=True)
    assert_parity(want, got, check_gp=False)


@pytest.mark.parametrize("every,cap", [(0, None), (3, None), (5, 16), (5, 256)])
@pytest.mark.parametrize("depth,N,adj,bins", [(20, 300, 0, False), (20, 130, 3, True), (70, 64, 0, False), (70, 64, 3, False), (12, 200, 1, False),
                                              (12, 200, 2, True), (100, 64, 0, False), (100, 70, 2, False)])
def test_deferred_build_with_the_optional_tags(oracle, monkeypatch, every, cap, depth, N, adj, bins):
    """k_sample<2, LEAN 3> (round 4): -addQS / -addI16 / strand tags / --adjust-qs run the deferred build too.  The dense pass gathers the
    owners' per-base quality sums with LDS atomics (--adjust-qs 0 / 3, at most 132 staged reads: cases 1-4, depth 70 = several pool
    segments, the sum words sharing LDS with the stream bases of the next segment), the owners' flush loop does otherwise (--adjust-qs 1 /
    2; depth 100: 196 staged reads); a read k_redo draws again is ADDED to the evaluation's quality sums and the site totals, which took
    0 for it (VGL_DEBUG_REDO_EVERY sends every k-th candidate there; cap 16: the list overflows into the bitmap)."""
    if every:
        monkeypatch.setenv("VGL_DEBUG_REDO_EVERY", str(every))
    if cap is not None:
        monkeypatch.setenv("VGL_DEBUG_REDO_CAP", str(cap))
    kw = dict(qs_bins=[(0, 2, 2), (3, 14, 12), (15, 30, 23), (31, 40, 37)]) if bins else {}
    args = VcfglArgs(seed=31, depth=depth, error_rate=0.01, error_qs=2, beta_variance=1e-5, adjust_qs=adj, **kw, **ALLTAGS, **STRAND)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    gt = synth.acgt_sites(9, N, seed=depth + adj, missing=0.03)
    sim = Simulator(args, N, device=0, max_sites_per_tile=9, hooks=True)
    info = sim.info()
    assert info["sample_lean"] == 3
    got = sim.simulate(0, gt)
    sim.close()
    want = oracle.Oracle(args, N).simulate(0, gt, fields=sim.default_fields())
    assert_parity(want, got, i16=True)


def test_optional_tags_keep_the_inline_build_for_dumps_and_small_shapes(oracle):
    """a per-read dump, or a beta shape parameter below 8, still runs k_sample<2, LEAN 0> (double-precision fallbacks inline)"""
    args = VcfglArgs(seed=31, depth=9, error_rate=0.2, error_qs=2, beta_variance=0.032, add_qs=1, add_i16=1)      # alpha = 0.8
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    sim = Simulator(args, 70, device=0, max_sites_per_tile=8)
    assert sim.info()["sample_lean"] == 0
    sim.close()
    want, got = run_both(oracle, args, synth.acgt_sites(8, 70, seed=2, missing=0.03))
    assert_parity(want, got, i16=True, check_gp=False)
    args = VcfglArgs(seed=31, depth=9, error_rate=0.01, error_qs=2, beta_variance=1e-5, add_qs=1, add_i16=1)
    want, got = run_both(oracle, args, synth.acgt_sites(8, 70, seed=2, missing=0.03), read_capacity=48)          # the dump: inline build
    assert np.array_equal(want.numpy("reads"), got.numpy("reads"))
    assert_parity(want, got, i16=True, check_gp=False)


@pytest.mark.parametrize("mean,var,depth", [(0.2, 0.032, 40), (0.05, 0.03, 120), (0.001, 5e-7, 25), (0.3, 0.15, 200), (0.02, 1e-4, 7)])
def test_gl1_per_read_scores_over_several_quality_windows(oracle, mean, var, depth):
    """k_gl<.,1> with per-read scores sorts each lane's reads by counting over windows of 16 quality values from the wavefront's
    top score down: beta shapes whose scores spread over the whole range [4, 63] (several windows per wavefront, bins with many
    reads, scores below errmod's clamp at 4), deep pileups (counts up to the depth), a narrow high-quality shape (one window)"""
    args = VcfglArgs(seed=17, depth=depth, error_rate=mean, error_qs=2, beta_variance=var, gl_model=1, add_pl=1, add_fmt_ad=1)
    want, got = run_both(oracle, args, synth.acgt_sites(8, 150, seed=21, missing=0.03), site0=40)
    assert_parity(want, got, check_gp=False)


@pytest.mark.parametrize("var", [1e-9, 1e-12])
def test_error_qs2_very_large_shape_parameters(oracle, var):
    """beta shape parameters of 1e4 .. 1e10 (--beta-variance 1e-9 / 1e-12): the gamma sampler's second test compares
    quantities of order x^4 / a1 against the rounding of a1 (1 - v + log v); the sure-accept bound of k_sample<2>
    carries a slack that grows with a1 (vgl_host.cpp: sure_margin)"""
    args = VcfglArgs(seed=9, depth=8, error_rate=0.01, error_qs=2, beta_variance=var, **ALLTAGS)
    want, got = run_both(oracle, args, synth.binary_sites(0, 40, 100), read_capacity=40)
    assert np.array_equal(want.numpy("reads"), got.numpy("reads"))
    assert_parity(want, got)


def test_rng_period_guard():
    """tile windows past the 2^48 period of rand48 are refused (VGL_E_ARG), the last admissible site is not"""
    import ctypes as C
    from vcfgl_amd.simulator import VglError
    args = VcfglArgs(seed=42, depth=30.0, error_rate=0.01, error_qs=2, beta_variance=1e-5)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    N = 2000
    sim = Simulator(args, N, device=0, max_sites_per_tile=4)
    mx = C.c_int64()
    assert sim.lib.vgl_rng_tile_max_sites(C.byref(sim.params), C.byref(mx)) == 0
    gt = synth.binary_sites(0, 2, N)
    sim.simulate(mx.value - 2, gt, fields=["fmt_dp"])                       # the last two sites of the period
    with pytest.raises(VglError) as e:
        sim.simulate(mx.value - 1, gt, fields=["fmt_dp"])
    assert e.value.code == _abi.VGL_E_ARG and "2^48" in str(e.value)
    sim.close()


@pytest.mark.parametrize("mode", [_abi.VGL_RNG_TILE, _abi.VGL_RNG_SERIAL])
@pytest.mark.parametrize("eqs", [0, 2])
def test_gl1_deeper_than_255_reads(oracle, mode, eqs):
    """GL model 1 at depth > 255 (--depth goes to 500): htslib's errmod_cal() shuffles the pileup with ks_shuffle on its own
    never-seeded rand48 stream (hts_drand48) and keeps the first 255 reads.  Samples below and above the limit mixed; in
    serial mode that stream runs on from evaluation to evaluation and from tile to tile."""
    depths = [300.0, 10.0, 262.0, 248.0, 420.0, 0.5, 256.0]
    N, S = len(depths), 11
    args = VcfglArgs(seed=42, depths=depths, error_rate=0.02, gl_model=1, error_qs=eqs, beta_variance=(1e-4 if eqs else -1.0),
                     add_pl=1, add_fmt_ad=1, rm_invar_sites=0)
    args.rng_mode = mode
    args.beta_sampler = _abi.VGL_BETA_STD if mode == _abi.VGL_RNG_SERIAL else _abi.VGL_BETA_RAND48
    gt = synth.acgt_sites(S, N, seed=5)
    fields = ["site_status", "n_alleles", "alleles2acgt", "fmt_dp", "fmt_ad", "gl", "pl"]
    want = oracle.Oracle(args, N).simulate(0, gt, fields=fields)
    sim = Simulator(args, N, device=0, max_sites_per_tile=4)
    parts = [sim.simulate(s0, gt[s0:s0 + 4], fields=fields) for s0 in range(0, S, 4)]          # three tiles
    sim.close()
    assert int(want.numpy("fmt_dp").max()) > 400 and int((want.numpy("fmt_dp") > 255).sum()) > 20
    for f in fields:
        got = np.concatenate([p.numpy(f) for p in parts], axis=0)
        w = want.numpy(f)
        assert np.array_equal(got.view(np.uint32) if got.dtype == np.float32 else got, w.view(np.uint32) if w.dtype == np.float32 else w), f


def test_async_host_path_two_tiles_in_flight(oracle):
    """vgl_simulate_tile_async / vgl_tile_wait: page-locked destination buffers, two tiles in flight -- the same
    bytes as the synchronous call and as the oracle; a third submission without a wait is refused, a qs-bin miss is reported
    by the wait of the tile it happened in."""
    import ctypes as C
    args = VcfglArgs(seed=42, depth=9, error_rate=0.01, error_qs=2, beta_variance=1e-5, add_pl=1)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    N, TS, n_tiles = 130, 64, 5
    sim = Simulator(args, N, device=0, max_sites_per_tile=TS)
    lib = sim.lib
    shapes = {"site_status": (TS,), "n_alleles": (TS,), "alleles2acgt": (TS, 5), "fmt_dp": (TS, N), "gl": (TS, sim.G, N), "pl": (TS, sim.G, N)}
    dt = {"site_status": np.int32, "n_alleles": np.int32, "alleles2acgt": np.int8, "fmt_dp": np.int32, "gl": np.float32, "pl": np.int32}
    sets = []
    for _ in range(2):
        t, arrs, ptrs = _abi.TileOut(), {}, []
        for f, shp in shapes.items():
            nbytes = int(np.prod(shp)) * np.dtype(dt[f]).itemsize
            p = lib.vgl_host_alloc(nbytes)
            assert p
            ptrs.append(p)
            arrs[f] = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,)).view(dt[f]).reshape(shp)
            setattr(t, f, p)
        sets.append((t, arrs, ptrs))
    gts = [synth.binary_sites(k * TS, TS, N) for k in range(n_tiles)]
    ticks = [C.c_int32(), C.c_int32()]
    got, pending = [], None
    for k in range(n_tiles):
        sim._check(lib.vgl_simulate_tile_async(sim.ctx, k * TS, TS, gts[k].ctypes.data, C.byref(sets[k & 1][0]), C.byref(ticks[k & 1])))
        if pending is not None:
            sim._check(lib.vgl_tile_wait(sim.ctx, ticks[pending & 1]))
            got.append({f: a.copy() for f, a in sets[pending & 1][1].items()})
        pending = k
    sim._check(lib.vgl_tile_wait(sim.ctx, ticks[pending & 1]))
    got.append({f: a.copy() for f, a in sets[pending & 1][1].items()})
    assert lib.vgl_tile_wait(sim.ctx, 0) == _abi.VGL_E_ARG      # nothing in flight any more
    t3 = [C.c_int32() for _ in range(3)]                        # two in flight is the limit: the third submission is refused
    assert lib.vgl_simulate_tile_async(sim.ctx, 0, TS, gts[0].ctypes.data, C.byref(sets[0][0]), C.byref(t3[0])) == 0
    assert lib.vgl_simulate_tile_async(sim.ctx, TS, TS, gts[1].ctypes.data, C.byref(sets[1][0]), C.byref(t3[1])) == 0
    assert lib.vgl_simulate_tile_async(sim.ctx, 2 * TS, TS, gts[2].ctypes.data, C.byref(sets[0][0]), C.byref(t3[2])) == _abi.VGL_E_ARG
    assert lib.vgl_tile_wait(sim.ctx, t3[0]) == 0 and lib.vgl_tile_wait(sim.ctx, t3[1]) == 0
    orc = oracle.Oracle(args, N)
    for k in range(n_tiles):
        want = orc.simulate(k * TS, gts[k], fields=list(shapes))
        sync = sim.simulate(k * TS, gts[k], fields=list(shapes))
        for f in shapes:
            w, g, s2 = want.numpy(f), got[k][f], sync.numpy(f)
            v = lambda x: x.view(np.uint32) if x.dtype == np.float32 else x
            assert np.array_equal(v(w), v(g)) and np.array_equal(v(w), v(s2)), (k, f)
    for _, _, ptrs in sets:
        for p in ptrs:
            lib.vgl_host_free(p)
    sim.close()
    # a tile whose simulated quality scores fall outside --qs-bins: reported by that tile's wait, and only by it
    bad = VcfglArgs(seed=42, depth=9, error_rate=0.01, error_qs=2, beta_variance=1e-5, qs_bins=[(0, 5, 3)])
    bad.rng_mode, bad.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    sim = Simulator(bad, N, device=0, max_sites_per_tile=TS)
    tile = sim.new_tile(TS, fields=["fmt_dp"])
    tk = C.c_int32()
    sim._check(lib.vgl_simulate_tile_async(sim.ctx, 0, TS, gts[0].ctypes.data, tile.byref(), C.byref(tk)))
    assert lib.vgl_tile_wait(sim.ctx, tk) == _abi.VGL_E_QSBIN
    sim.close()


# ---- ABI 4: sample-major FORMAT arrays (what simRecord::add_tags() hands to bcf_update_format_*) and PL in one byte

def _layout_case(oracle, N, n_sites, kw, site0=5):
    args = VcfglArgs(seed=42, error_rate=0.02, **kw)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    gt = synth.acgt_sites(n_sites, N, seed=N, missing=0.05)
    out = {}
    for lay in (_abi.VGL_LAYOUT_PLANES, _abi.VGL_LAYOUT_SAMPLE_MAJOR):
        args.out_layout = lay
        sim = Simulator(args, N, device=0, max_sites_per_tile=n_sites)
        fields = sim.default_fields() + ["pl_u8"]
        out[lay] = (oracle.Oracle(args, N).simulate(site0, gt, fields=fields), sim.simulate(site0, gt, fields=fields), fields)
        sim.close()
    return out


@pytest.mark.parametrize("N,n_sites", [(1, 9), (2, 9), (63, 7), (64, 7), (65, 7), (130, 12), (1000, 5), (1027, 3)])
@pytest.mark.parametrize("kw", [dict(depth=6.0, do_unobserved=1, rm_invar_sites=4, rm_empty_sites=1, **ALLTAGS, **STRAND),
                                dict(depth=1.0, do_unobserved=0, **ALLTAGS),
                                dict(depth=14.0, do_unobserved=5, error_qs=2, beta_variance=1e-4, add_pl=1, add_fmt_ad=1),
                                dict(depth=9.0, do_unobserved=3, gl_model=1, add_pl=1, add_fmt_ad=1)])
def test_sample_major_layout_and_narrow_pl(oracle, N, n_sites, kw):
    """VGL_LAYOUT_SAMPLE_MAJOR: the slab of a site holds x[sample * nK + k] with the site's own nK (bcf_utils.h:193-196) -- equal
    to the oracle's slabs byte for byte (nothing else of a slab is written), and to the planes of the same run read the other
    way round; pl_u8 = PL in one byte, 255 where PL is missing.  Ragged sample counts: partial last wavefronts, sites of 1 .. 5
    alleles, skipped and empty sites."""
    out = _layout_case(oracle, N, n_sites, kw)
    P, SM = _abi.VGL_LAYOUT_PLANES, _abi.VGL_LAYOUT_SAMPLE_MAJOR
    MULTI = {"gl": "G", "pl": "G", "gp": "G", "pl_u8": "G", "fmt_ad": "A", "fmt_adf": "A", "fmt_adr": "A"}

    def same(f, a, b, tag):
        if f == "gp":                                                    # device 10^x: 1e-6 absolute where the value is not missing
            m = a.view(np.uint32) == _abi.FLOAT_MISSING_BITS
            assert np.array_equal(m, b.view(np.uint32) == _abi.FLOAT_MISSING_BITS), tag
            assert np.all(np.abs(a[~m].astype(np.float64) - b[~m].astype(np.float64)) <= TOL), tag
        else:
            assert np.array_equal(bits(np.ascontiguousarray(a)), bits(np.ascontiguousarray(b))), tag

    for lay in (P, SM):
        want, got, fields = out[lay]
        st, na = want.numpy("site_status"), want.numpy("n_alleles")
        for f in fields:
            a, b = want.numpy(f), got.numpy(f)
            if f == "i16":
                a, b = a[:, :12], b[:, :12]
            if lay == SM and f in MULTI:
                # the record arrays themselves (what lies behind them in a slab is unspecified: the host entry points copy the whole
                # slab back, the kernels write only the record's n_samples x nK values)
                for i in range(n_sites):
                    nA = int(na[i]) if st[i] >= 0 else 0
                    nk = nA * (nA + 1) // 2 if MULTI[f] == "G" else nA
                    same(f, want.site_records(f, i, nk), got.site_records(f, i, nk), (lay, f, i))
            else:
                same(f, a, b, (lay, f))
    planes, slabs = out[P][1], out[SM][1]
    st, na = planes.numpy("site_status"), planes.numpy("n_alleles")
    for i in range(n_sites):
        if st[i] < 0:
            continue
        nA = int(na[i]); nG = nA * (nA + 1) // 2
        for f, nk in (("gl", nG), ("pl", nG), ("pl_u8", nG), ("fmt_ad", nA), ("fmt_adf", nA), ("fmt_adr", nA)):
            if f not in planes.arrays:
                continue
            assert np.array_equal(bits(slabs.site_records(f, i, nk)), bits(np.ascontiguousarray(planes.numpy(f)[i, :nk, :].T))), (i, f)
    pl, u8 = planes.numpy("pl"), planes.numpy("pl_u8")
    assert np.array_equal(u8, np.where(pl == _abi.INT32_MISSING, 255, pl).astype(np.uint8))


def test_async_tile_that_fails_after_its_first_enqueue_leaves_the_context_usable(oracle):
    """vgl_simulate_tile_async commits ticket and slot only when the whole tile is enqueued: a call that fails part-way (here: an
    INFO/QS buffer without -addQS, refused after the genotypes' copy was enqueued) drains the streams, clears the device error
    word and frees the slot; the next tiles run, two in flight, with the tickets and the values of an undisturbed context"""
    import ctypes as C
    args = VcfglArgs(seed=42, depth=12.0, error_rate=0.01, add_pl=1)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    N, S = 70, 16
    sim = Simulator(args, N, device=0, max_sites_per_tile=S)
    gts = [synth.binary_sites(k * S, S, N) for k in range(3)]
    bad = sim.new_tile(S, fields=["fmt_dp", "gl", "qs"])                 # qs needs -addQS in the context parameters
    t = C.c_int32(-7)
    rc = sim.lib.vgl_simulate_tile_async(sim.ctx, 0, S, gts[0].ctypes.data, bad.byref(), C.byref(t))
    assert rc == _abi.VGL_E_ARG and t.value == -7
    assert sim.lib.vgl_tile_wait(sim.ctx, 0) == _abi.VGL_E_ARG           # nothing is in flight under either ticket
    tiles, tick = [sim.new_tile(S, fields=["fmt_dp", "gl", "pl"]) for _ in range(3)], [C.c_int32() for _ in range(3)]
    sim._check(sim.lib.vgl_simulate_tile_async(sim.ctx, 0, S, gts[0].ctypes.data, tiles[0].byref(), C.byref(tick[0])))
    sim._check(sim.lib.vgl_simulate_tile_async(sim.ctx, S, S, gts[1].ctypes.data, tiles[1].byref(), C.byref(tick[1])))
    assert (tick[0].value, tick[1].value) == (0, 1)
    sim._check(sim.lib.vgl_tile_wait(sim.ctx, tick[0]))
    sim._check(sim.lib.vgl_simulate_tile_async(sim.ctx, 2 * S, S, gts[2].ctypes.data, tiles[2].byref(), C.byref(tick[2])))
    sim._check(sim.lib.vgl_tile_wait(sim.ctx, tick[1]))
    sim._check(sim.lib.vgl_tile_wait(sim.ctx, tick[2]))
    sim.close()
    orc = oracle.Oracle(args, N)
    for k in range(3):
        want = orc.simulate(k * S, gts[k], fields=["fmt_dp", "gl", "pl"])
        for f in ("fmt_dp", "pl", "gl"):
            assert np.array_equal(bits(want.numpy(f)), bits(tiles[k].numpy(f))), (k, f)


@pytest.mark.parametrize("depth,tags,lean", [(140, {}, 2), (140, dict(add_qs=1, add_i16=1), 3), (150, {}, 1), (150, dict(add_qs=1), 0)])
def test_two_byte_items_at_the_edge_of_their_read_index(oracle, depth, tags, lean):
    """round 5: the float32 builds of k_sample<2> keep a work item in two bytes with 8 bits of read index -- mean depth 140 stages up to
    251 reads (still those builds, several pool segments), depth 150 up to 264: the inline-fallback build (LEAN 0 / 1) takes over.
    Both equal to the oracle."""
    args = VcfglArgs(seed=5, depth=depth, error_rate=0.01, error_qs=2, beta_variance=1e-5, add_pl=1, **tags)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    sim = Simulator(args, 70, device=0, max_sites_per_tile=6)
    info = sim.info()
    sim.close()
    assert info["sample_lean"] == lean and (info["read_cap"] <= 256) == (lean >= 2), info
    want, got = run_both(oracle, args, synth.binary_sites(0, 6, 70))
    assert_parity(want, got, check_gp=False)


@pytest.mark.parametrize("bins,lean", [([(0, 20, 10), (21, 254, 30)], 2), ([(0, 20, 10), (21, 300, 30)], 1)])
def test_binned_scores_from_the_lds_table_and_beyond_it(oracle, bins, lean):
    """--qs-bins in the two-byte-item builds: the finishing lane looks the binned score up in a 256-entry LDS table; a bin that reaches
    beyond 254 cannot be tabulated and sends the run to the inline build (vgl_ctx_create)."""
    args = VcfglArgs(seed=9, depth=12, error_rate=0.02, error_qs=2, beta_variance=1e-5, qs_bins=bins, add_pl=1)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    sim = Simulator(args, 130, device=0, max_sites_per_tile=12)
    info = sim.info()
    sim.close()
    assert info["sample_lean"] == lean, info
    want, got = run_both(oracle, args, synth.binary_sites(0, 12, 130))
    assert_parity(want, got, check_gp=False)
