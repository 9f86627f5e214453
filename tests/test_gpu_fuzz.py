"""Randomised differential test: random flag combinations, shapes and inputs; the HIP path (tile mode
and serial mode) against the CPU oracle on every output field."""
import os

import numpy as np
import pytest

import synth
from vcfgl_amd import Simulator, VcfglArgs, VcfglArgError, _abi

pytestmark = pytest.mark.gpu
INT_FIELDS = ["site_status", "n_alleles", "n_alleles_obs", "alleles2acgt", "info_dp", "info_ad", "info_adf",
              "info_adr", "fmt_dp", "pl", "fmt_ad", "fmt_adf", "fmt_adr"]


def random_case(rng):
    eqs = int(rng.choice([0, 0, 1, 2, 2]))
    gl = int(rng.choice([1, 2, 2]))
    precise = int(rng.integers(0, 2)) if gl == 2 else 0
    adj = int(rng.choice([0, 0, 1, 2, 3])) if not precise else int(rng.choice([0, 2]))
    add_qs = 1 if (adj & 2) else int(rng.integers(0, 2))
    strand = int(rng.integers(0, 2))
    kw = dict(
        seed=int(rng.integers(-2 ** 31, 2 ** 31 - 1)),
        error_rate=float(rng.choice([0.0, 0.001, 0.01, 0.05, 0.3])) if eqs == 0 else float(rng.choice([0.005, 0.01, 0.05, 0.2])),
        error_qs=eqs, gl_model=gl, precise_gl=precise, adjust_qs=adj,
        adjust_by=float(rng.choice([0.499, 0.25, 1.0])),
        do_unobserved=int(rng.integers(0, 6)), rm_invar_sites=int(rng.choice([0, 0, 4])), rm_empty_sites=int(rng.integers(0, 2)),
        add_pl=int(rng.integers(0, 2)), add_gp=int(rng.integers(0, 2)), add_qs=add_qs, add_info_dp=1, add_fmt_ad=1, add_info_ad=1,
        add_i16=strand & int(rng.integers(0, 2)), add_fmt_adf=strand, add_fmt_adr=strand, add_info_adf=strand, add_info_adr=strand,
        gl1_theta=float(rng.choice([0.83, 0.5])), i16_mapq=20,
    )
    if eqs:
        m = kw["error_rate"]
        kw["beta_variance"] = float(rng.choice([1e-5, 1e-4, m * (1 - m) * 0.2]))
    N = int(rng.choice([1, 3, 17, 64, 65, 130, 300]))
    if rng.random() < 0.25:
        kw["depths"] = [float(x) for x in rng.choice([0.0, 0.3, 2.0, 8.0, 13.0, 25.0, 290.0], size=N, p=[0.18, 0.18, 0.18, 0.18, 0.12, 0.12, 0.04])]
    else:
        kw["depth"] = float(rng.choice([0.0, 0.2, 1.0, 4.0, 11.0, 12.0, 18.0, 33.0, 262.0], p=[0.12] * 8 + [0.04]))   # 262: GL model 1 beyond errmod's 255 reads
    if eqs == 2 and rng.random() < 0.3:
        kw["qs_bins"] = [(0, 2, 2), (3, 14, 12), (15, 30, 23), (31, 63, 37)]
    S = int(rng.integers(1, 40))
    gt = synth.acgt_sites(S, N, seed=int(rng.integers(0, 1 << 30)), missing=float(rng.choice([0.0, 0.0, 0.05, 0.5])),
                          n_alleles=int(rng.integers(1, 5)))
    return kw, N, gt


def compare(want, got, tol_gl, tag):
    for f in INT_FIELDS:
        assert np.array_equal(want.numpy(f), got.numpy(f)), (tag, f)
    wb, gb = want.numpy("gl").view(np.uint32), got.numpy("gl").view(np.uint32)
    if not tol_gl:
        assert np.array_equal(wb, gb), (tag, "gl", int(np.sum(wb != gb)))
    else:
        miss = wb == _abi.FLOAT_MISSING_BITS
        assert np.array_equal(miss, gb == _abi.FLOAT_MISSING_BITS), (tag, "gl missing")
        # a device log10 feeds these values (--precise-gl 1): at most 1 unit in the last place of float32 from the oracle's
        from test_gpu_parity import ulps32
        a, b = want.numpy("gl")[~miss], got.numpy("gl")[~miss]
        fin = np.isfinite(a)
        assert np.array_equal(fin, np.isfinite(b)) and np.array_equal(a[~fin], b[~fin]), (tag, "gl non-finite")
        d = ulps32(a[fin], b[fin])
        assert d.size == 0 or (d.max() <= 1 and (d > 0).sum() <= max(1, int(np.ceil(1e-5 * d.size)))), (tag, "gl ulp", int(d.max()), int((d > 0).sum()))
    if "gp" in got.arrays:
        m = want.numpy("gp").view(np.uint32) == _abi.FLOAT_MISSING_BITS
        assert np.array_equal(m, got.numpy("gp").view(np.uint32) == _abi.FLOAT_MISSING_BITS), (tag, "gp missing")
        d = np.abs(want.numpy("gp")[~m].astype(np.float64) - got.numpy("gp")[~m].astype(np.float64))
        assert np.all(~(d > 1e-6)), (tag, "gp")
    if "qs" in got.arrays:
        assert np.array_equal(want.numpy("qs").view(np.uint32), got.numpy("qs").view(np.uint32)), (tag, "qs")
    if "i16" in got.arrays:
        assert np.array_equal(want.numpy("i16").view(np.uint32), got.numpy("i16").view(np.uint32)), (tag, "i16")


# VGL_FUZZ_CHUNKS / VGL_FUZZ_SEED: longer one-off runs (10 configurations x 2 RNG modes per chunk)
@pytest.mark.parametrize("chunk", range(int(os.environ.get("VGL_FUZZ_CHUNKS", "24"))))
def test_random_configurations(oracle, chunk):
    rng = np.random.default_rng(int(os.environ.get("VGL_FUZZ_SEED", "1000")) + chunk)
    done = 0
    while done < 10:
        kw, N, gt = random_case(rng)
        try:
            base = VcfglArgs(**kw).validate()
        except VcfglArgError:
            continue
        for mode, beta in ((_abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48), (_abi.VGL_RNG_SERIAL, _abi.VGL_BETA_STD)):
            args = VcfglArgs(**kw)
            args.rng_mode, args.beta_sampler = mode, beta
            tag = (chunk, done, mode, kw, N, gt.shape)
            try:
                orc = oracle.Oracle(args, N)
            except oracle.OracleError:
                break                                   # e.g. non-positive beta shape parameters: rejected by both
            sim = Simulator(args, N, max_sites_per_tile=gt.shape[0])
            fields = sim.default_fields()
            if not args.add_pl:
                fields = [f for f in fields if f != "pl"] + ["pl"]
            try:
                want = orc.simulate(0, gt, fields=fields)
            except oracle.OracleError as e:
                with pytest.raises(Exception):
                    sim.simulate(0, gt, fields=fields)  # same refusal (qs-bin miss, GL1 depth > 255)
                sim.close()
                continue
            got = sim.simulate(0, gt, fields=fields)
            sim.close()
            compare(want, got, tol_gl=bool(args.precise_gl), tag=tag)
        done += 1


def test_error_probability_zero_with_adjusted_scores_is_refused(oracle):
    """Found by a longer run of the test above (VGL_FUZZ_SEED=50000, chunk 74, configuration 7): beta(0.02, 3.98) quality-score
    noise gives a read an error probability of exactly 0, for which the reference leaves the adjusted quality score
    at -1 and then exits on ASSERT(adjqScore_i != -1) (vcfgl.cpp:558; gl_methods.cpp:101 for --adjust-qs 1).  Serial
    mode reaches that read: oracle and HIP path both refuse with VGL_E_ADJQ; without --adjust-qs the run is fine."""
    kw = {'seed': -548263391, 'error_rate': 0.005, 'error_qs': 2, 'gl_model': 2, 'precise_gl': 0, 'adjust_qs': 3, 'adjust_by': 0.25,
          'do_unobserved': 0, 'rm_invar_sites': 0, 'rm_empty_sites': 0, 'add_pl': 0, 'add_gp': 0, 'add_qs': 1, 'add_info_dp': 1, 'add_fmt_ad': 1,
          'add_info_ad': 1, 'add_i16': 0, 'add_fmt_adf': 0, 'add_fmt_adr': 0, 'add_info_adf': 0, 'add_info_adr': 0, 'gl1_theta': 0.83, 'i16_mapq': 20,
          'beta_variance': 0.000995, 'depth': 1.0}                     # the configuration that run generated, written out
    N = 300
    gt = synth.acgt_sites(29, N, seed=942061679, missing=0.0, n_alleles=3)
    for adj, fails in ((3, True), (0, False)):
        args = VcfglArgs(**dict(kw, adjust_qs=adj))
        args.rng_mode, args.beta_sampler = _abi.VGL_RNG_SERIAL, _abi.VGL_BETA_STD
        sim = Simulator(args, N, max_sites_per_tile=gt.shape[0])
        orc = oracle.Oracle(args, N)
        if fails:
            with pytest.raises(oracle.OracleError) as eo:
                orc.simulate(0, gt, fields=sim.default_fields())
            assert eo.value.code == _abi.VGL_E_ADJQ
            with pytest.raises(Exception) as ei:
                sim.simulate(0, gt)
            assert getattr(ei.value, "code", None) == _abi.VGL_E_ADJQ
        else:
            compare(orc.simulate(0, gt, fields=sim.default_fields()), sim.simulate(0, gt), tol_gl=False, tag="adj0")
        sim.close()
