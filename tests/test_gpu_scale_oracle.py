"""Oracle parity at scale: 1e7 - 1e8 evaluations per case compared with the CPU oracle FIELD BY FIELD and SITE BY SITE (one
64-bit checksum per site and field on both sides, tests/oracle_pool.py), not by sampling -- one case per BUILD of the kernels,
so that the rare paths of each (the reads a float32 bound cannot settle, ~6 in 10^4; three / four distinct bases; overflowing
pools) are reached through volume, not only through debug hooks:

  c3       BASELINE.json configs[2] flags, 100,000 sites x 1000 samples = 1e8 evaluations: k_sample<2, LEAN 2> + k_redo, k_gl<5,2>
  c4       configs[3] flags (depth 30, rta3 qs-bins), 5,000 x 2000 = 1e7: two pool segments per wavefront
  c5       configs[4] flags (exploded hom-ref sites, -doUnobserved 2, PL), 200,000 x 500 = 1e8: k_sample<0>, product-method depths
  c2       configs[1] AT FULL SIZE, 10,000 x 100, depth 10, -GL 1: k_gl<.,1> on table lookups
  alltags  every optional tag incl. strand tags and I16, --error-qs 2, multi-allelic input with missing calls, 1e7:
           k_sample<2, LEAN 0> (inline double fallbacks), k_siteagg, ADF / ADR / QS / I16[0..11], GP
  adjbins  --adjust-qs 3 + --qs-bins + -addQS, 1e7: adjusted scores through the bins, into GL and the quality sums
  precise  --precise-gl 1, 1e7: k_sample<2, PREC>, k_gl<5,2,PREC>; GL compared in units in the last place (below)
  gl1q     -GL 1 --error-qs 2, 1e7: k_gl<.,1> with per-read scores
  eqs1     --error-qs 1 (one beta deviate per site), 1e7
  fixedq   --error-qs 0 with strand tags, 1e7: k_sample<0> with strand draws, k_depth
  c3sm     c3's flags + PL, VGL_LAYOUT_SAMPLE_MAJOR slabs and pl_u8 (ABI 4), multi-allelic input, 1e7: k_gl's sample-major stores
  qsi16    (round 4) c3's flags + -addQS -addI16, 1e7: k_sample<2, LEAN 3> (quality sums by LDS atomics, k_redo adding to them), k_siteagg
  fq20     (round 4) one fixed score at depth 20, default tags + PL, 2e7: sample_reads_fixed (haplotype bits, homozygous wavefronts), the
           one-base run table of k_gl
  c5wide   (round 4) config C5's flags at 1000 samples, 2e7: the fused kernel split over two workgroups per site

The site ranges start far from 0 (absolute site indexing: the same values the full job produces there).  Integer fields, and GL
wherever its per-read terms come from the qScore LUT or constants, must be EQUAL.  Where the device evaluates a logarithm or a
power itself (--precise-gl 1: log10_unit; GP: exp10_nonpos; both ~1e-16 relative against glibc's), a float32 result can land on
the other side of a rounding boundary: those fields are compared value by value on the sites whose checksums differ -- at most
1 unit in the last place of float32 for GL and 1e-6 absolute for GP (`north_star`'s tolerance), the number of non-identical
values printed and bounded by 1e-5 of all values."""
import dataclasses

import numpy as np
import pytest
import torch

import oracle_pool
import synth
from vcfgl_amd import Simulator, VcfglArgs, _abi

pytestmark = pytest.mark.gpu

RTA3 = [(0, 2, 2), (3, 14, 12), (15, 30, 23), (31, 40, 37)]
BASE = ["site_status", "n_alleles", "alleles2acgt", "fmt_dp", "gl"]
EVERY = ["site_status", "n_alleles", "n_alleles_obs", "alleles2acgt", "info_dp", "info_ad", "info_adf", "info_adr", "qs", "i16",
         "fmt_dp", "gl", "pl", "gp", "fmt_ad", "fmt_adf", "fmt_adr"]
ALLTAGS = dict(add_gp=1, add_pl=1, add_qs=1, add_info_dp=1, add_fmt_ad=1, add_info_ad=1)
STRAND = dict(add_i16=1, add_fmt_adf=1, add_info_adf=1, add_fmt_adr=1, add_info_adr=1)
EQS2 = dict(error_rate=0.01, error_qs=2, beta_variance=1e-5)
CASES = {
    "c3": dict(N=1000, S=100_000, site0=400_000, gt="binary", fields=BASE, flags=dict(depth=20.0, gl_model=2, **EQS2)),
    "c4": dict(N=2000, S=5_000, site0=9_000_000, gt="binary", fields=BASE, flags=dict(depth=30.0, gl_model=2, qs_bins=RTA3, **EQS2)),
    "c5": dict(N=500, S=200_000, site0=40_000_000, gt="homref", fields=BASE + ["pl"],
               flags=dict(depth=5.0, error_rate=0.01, gl_model=2, do_unobserved=2, add_pl=1)),
    "c2": dict(N=100, S=10_000, site0=0, gt="binary", fields=BASE + ["pl", "fmt_ad"], tile=4096,
               flags=dict(depth=10.0, error_rate=0.01, gl_model=1, add_pl=1, add_fmt_ad=1)),
    "alltags": dict(N=1000, S=10_000, site0=3_000_000, gt="acgt", fields=EVERY, tile=4096, soft=("gp",),
                    flags=dict(depth=20.0, gl_model=2, **EQS2, **ALLTAGS, **STRAND)),
    "adjbins": dict(N=1000, S=10_000, site0=5_000_000, gt="binary", fields=BASE + ["pl", "qs", "fmt_ad"], tile=4096,
                    flags=dict(depth=20.0, gl_model=2, adjust_qs=3, qs_bins=RTA3, add_pl=1, add_qs=1, add_fmt_ad=1, **EQS2)),
    "precise": dict(N=1000, S=10_000, site0=7_000_000, gt="binary", fields=BASE + ["pl"], tile=4096, soft=("gl",),
                    flags=dict(depth=20.0, gl_model=2, precise_gl=1, add_pl=1, **EQS2)),
    "gl1q": dict(N=1000, S=10_000, site0=11_000_000, gt="binary", fields=BASE + ["pl"], tile=4096,
                 flags=dict(depth=20.0, gl_model=1, add_pl=1, **EQS2)),
    "eqs1": dict(N=1000, S=10_000, site0=13_000_000, gt="binary", fields=BASE + ["pl", "fmt_ad"], tile=4096,
                 flags=dict(depth=20.0, gl_model=2, error_rate=0.01, error_qs=1, beta_variance=1e-5, add_pl=1, add_fmt_ad=1)),
    "c3sm": dict(N=1000, S=10_000, site0=17_000_000, gt="acgt", fields=BASE + ["pl", "pl_u8", "fmt_ad"], tile=4096,
                 flags=dict(depth=20.0, gl_model=2, add_pl=1, add_fmt_ad=1, out_layout=_abi.VGL_LAYOUT_SAMPLE_MAJOR, **EQS2)),
    "fixedq": dict(N=1000, S=10_000, site0=15_000_000, gt="acgt", fields=[f for f in EVERY if f != "gp"], tile=4096,
                   flags=dict(depth=20.0, gl_model=2, error_rate=0.01, add_pl=1, add_qs=1, add_info_dp=1, add_fmt_ad=1, add_info_ad=1, **STRAND)),
    # round 4
    "qsi16": dict(N=1000, S=10_000, site0=19_000_000, gt="binary", fields=BASE + ["qs", "i16"], tile=4096,
                  flags=dict(depth=20.0, gl_model=2, add_qs=1, add_i16=1, **EQS2)),
    "fq20": dict(N=1000, S=20_000, site0=21_000_000, gt="binary", fields=BASE + ["pl"], tile=8192,
                 flags=dict(depth=20.0, gl_model=2, error_rate=0.01, add_pl=1)),
    "c5wide": dict(N=1000, S=20_000, site0=23_000_000, gt="homref", fields=BASE + ["pl"], tile=8192,
                   flags=dict(depth=5.0, error_rate=0.01, gl_model=2, do_unobserved=2, add_pl=1)),
}


def _args(flags):
    a = VcfglArgs(seed=42, **flags)
    a.rng_mode, a.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    return a


def _gt_torch(case, site0, n, dev):
    N = case["N"]
    if case["gt"] == "homref":
        return torch.zeros((n, N), dtype=torch.uint8, device=dev)
    if case["gt"] == "acgt":
        return torch.from_numpy(synth.acgt_range(site0, n, N)).to(dev)
    return synth.binary_sites_torch(site0, n, N, dev)


def _gpu_checksums(case):
    dev = torch.device("cuda", 0)
    N, S, site0, tile_sites = case["N"], case["S"], case["site0"], case.get("tile", 8192)
    sim = Simulator(_args(case["flags"]), N, max_sites_per_tile=tile_sites)
    tile = sim.new_tile(tile_sites, fields=case["fields"], device=dev)
    out = np.zeros((S, len(case["fields"])), dtype=np.uint64)
    for s0 in range(0, S, tile_sites):
        n = min(tile_sites, S - s0)
        if case["flags"].get("out_layout"):                     # sample-major slabs: what lies behind a record's array is not written
            for f in case["fields"]:
                tile[f].zero_()
        sim.simulate_device(site0 + s0, _gt_torch(case, site0 + s0, n, dev), tile)
        sim.check()
        for k, f in enumerate(case["fields"]):
            out[s0:s0 + n, k] = oracle_pool.site_checksums_torch(oracle_pool.field_view(f, tile[f][:n]))
    return out, sim


def _ulps32(a, b):
    """distance of two finite float32 arrays in units in the last place (monotone integer image of the floats)"""
    def key(x):
        i = x.view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)
    return np.abs(key(a) - key(b))


def _recheck_soft(oracle, case, sim, field, sites):
    """value-by-value comparison of a float field on the given absolute sites: (non-identical values, worst distance)"""
    N = case["N"]
    orc = oracle.Oracle(_args(case["flags"]), N)
    nonid, worst = 0, 0.0
    for site in sites:
        gt = _gt_torch(case, site, 1, "cpu").numpy()
        g, w = sim.simulate(site, gt, fields=[field]).numpy(field), orc.simulate(site, gt, fields=[field]).numpy(field)
        gb, wb = g.view(np.uint32), w.view(np.uint32)
        miss = wb == _abi.FLOAT_MISSING_BITS
        assert np.array_equal(miss, gb == _abi.FLOAT_MISSING_BITS), (field, site, "missing pattern")
        fin = np.isfinite(w) & ~miss
        assert np.array_equal(gb[~fin], wb[~fin]), (field, site, "non-finite values")
        d = _ulps32(g[fin], w[fin]) if field == "gl" else np.abs(g[fin].astype(np.float64) - w[fin].astype(np.float64))
        nonid += int((gb != wb).sum())
        worst = max(worst, float(d.max()) if d.size else 0.0)
    return nonid, worst


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", list(CASES))
def test_every_site_equals_oracle(oracle, name):
    case = CASES[name]
    got, sim = _gpu_checksums(case)
    d = dataclasses.asdict(_args(case["flags"]))
    want = oracle_pool.oracle_site_checksums(d, case["N"], case["site0"], case["S"], case["fields"], gt=case["gt"])
    assert got.shape == want.shape == (case["S"], len(case["fields"]))
    soft = case.get("soft", ())
    notes = []
    for k, f in enumerate(case["fields"]):
        bad = np.flatnonzero(got[:, k] != want[:, k])
        if f not in soft:
            assert bad.size == 0, f"{name}: {bad.size} sites differ in {f}; first: site {case['site0'] + (bad[0] if bad.size else 0)}"
            continue
        per_site = case["N"] * sim.G
        assert bad.size <= max(20, 1e-5 * case["S"] * per_site), f"{name}: {bad.size} sites differ in {f}"
        nonid, worst = _recheck_soft(oracle, case, sim, f, [case["site0"] + int(b) for b in bad])
        assert nonid <= 1e-5 * case["S"] * per_site, (name, f, nonid)
        if f == "gl":
            assert worst <= 1, f"{name}: GL {worst} units in the last place from the oracle"
        else:
            assert worst <= 1e-6, f"{name}: {f} {worst} from the oracle"
        notes.append(f"{f}: {nonid} of {case['S'] * per_site:.3g} values not identical (worst {worst:g} {'ulp' if f == 'gl' else 'absolute'})")
    sim.close()
    print(f"{name}: {case['S'] * case['N']:.3g} evaluations, {case['S']} sites x {len(case['fields'])} fields equal to the oracle" +
          ("; " + "; ".join(notes) if notes else ""))
