"""Shapes at the edges of what the flags allow: 70 000 samples per site (1094 wavefronts per site, a ragged last one), one
sample and 20 000 sites (k_depth chunks spanning 1024 sites), --depth 499 (io.cpp:861 allows up to 500; staging capacity 695 reads:
GL model 1 subsamples to 255, GL model 2 runs several pool segments per wavefront), depth 0 (every site without reads)."""
import numpy as np
import pytest

import synth
from vcfgl_amd import Simulator, VcfglArgs, _abi

pytestmark = pytest.mark.gpu
FIELDS = ["site_status", "n_alleles", "alleles2acgt", "fmt_dp", "fmt_ad", "pl", "gl"]


@pytest.mark.parametrize("N,S,site0,kw", [
    (70000, 2, 0, dict(depth=3.0)), (70001, 3, 12345, dict(depth=14.0)), (5, 7, 0, dict(depth=499.0, gl_model=1)),
    (200, 3, 0, dict(depth=499.0, error_qs=2, beta_variance=1e-5)), (1000, 2, 0, dict(depth=499.0)),
    (3, 4000, 0, dict(depth=25.0, error_qs=2, beta_variance=1e-5)), (1, 20000, 0, dict(depth=13.0)), (64, 1, 0, dict(depth=0.0))])
def test_extreme_shapes_equal_the_oracle(oracle, N, S, site0, kw):
    a = VcfglArgs(seed=5, error_rate=0.01, add_pl=1, add_fmt_ad=1, **kw)
    a.rng_mode, a.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    gt = synth.binary_sites(site0, S, N)
    sim = Simulator(a, N, max_sites_per_tile=S)
    got = sim.simulate(site0, gt, fields=FIELDS)
    sim.close()
    want = oracle.Oracle(a, N).simulate(site0, gt, fields=FIELDS)
    for f in FIELDS:
        x, y = got.numpy(f), want.numpy(f)
        if x.dtype == np.float32:
            x, y = x.view(np.uint32), y.view(np.uint32)
        assert np.array_equal(x, y), f
