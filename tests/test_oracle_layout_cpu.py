"""The oracle's own VGL_LAYOUT_SAMPLE_MAJOR write-out (include/vcfgl_hip.h) against its planes: the slab of a site is the
planes of the same run read the other way round (x[sample * nK + k]), nothing else of the slab is written; pl_u8 = PL in a byte."""
import numpy as np
import pytest

import synth
from vcfgl_amd import VcfglArgs, _abi


@pytest.mark.parametrize("du", [0, 1, 3, 5])
def test_slabs_are_the_planes_transposed(oracle, du):
    N, S = 37, 60
    gt = synth.acgt_sites(S, N, seed=3 + du, missing=0.05)
    res = {}
    for lay in (_abi.VGL_LAYOUT_PLANES, _abi.VGL_LAYOUT_SAMPLE_MAJOR):
        a = VcfglArgs(seed=42, depth=0.08, error_rate=0.25, do_unobserved=du, rm_invar_sites=4, rm_empty_sites=1, add_pl=1, add_gp=1,
                      add_fmt_ad=1, out_layout=lay)
        a.rng_mode = _abi.VGL_RNG_TILE
        res[lay] = oracle.Oracle(a, N).simulate(0, gt, fields=["gl", "pl", "gp", "fmt_ad", "pl_u8", "fmt_dp"])
    planes, slabs = res[_abi.VGL_LAYOUT_PLANES], res[_abi.VGL_LAYOUT_SAMPLE_MAJOR]
    st, na = planes.numpy("site_status"), planes.numpy("n_alleles")
    assert (st < 0).any() and (st >= 0).any()
    for i in range(S):
        nA = int(na[i]) if st[i] >= 0 else 0
        nG = nA * (nA + 1) // 2
        for f, nk in (("gl", nG), ("pl", nG), ("gp", nG), ("pl_u8", nG), ("fmt_ad", nA)):
            x, y = slabs.site_records(f, i, nk), planes.numpy(f)[i, :nk, :].T
            if x.dtype == np.float32:
                x, y = x.view(np.uint32), np.ascontiguousarray(y).view(np.uint32)
            assert np.array_equal(x, y), (i, f)
            rest = slabs.numpy(f)[i].reshape(-1)[N * nk:]
            assert not rest.any(), (i, f, "written beyond the record's array")
    pl, u8 = planes.numpy("pl"), planes.numpy("pl_u8")
    assert np.array_equal(u8, np.where(pl == _abi.INT32_MISSING, 255, pl).astype(np.uint8))
