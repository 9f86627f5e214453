"""Resource budget of every kernel instantiation of the shipped library.

hipcc cross-compiles gfx950 without a GPU: the device code of each kernel file is compiled with the Makefile's own flags and the
compiler's kernel-resource-usage remarks are compared with profiles/kernel_budget.json.  A change that costs ANY shipped build registers,
scalar or vector spills, scratch, LDS or wavefronts per SIMD fails here -- round 5's 4 % regression of the depth-5 configuration came from a
two-line edit of a test hook in code that configuration never runs (51 scalar spills in the fused kernel) and no test saw it.
An intended change updates the file: `python tools/kernel_budget.py --update` (and the diff of the numbers goes into the commit)."""
import json
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_budget  # noqa: E402


@pytest.fixture(scope="module")
def current(tmp_path_factory):
    hipcc = kernel_budget.makefile_flags()[0]
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    return kernel_budget.current(outdir=str(tmp_path_factory.mktemp("budget")))


def test_parse_both_remark_layouts():
    a = "remark: vgl_sample.hip:1026:0: Function Name: _Zfoo [-Rpass-analysis=kernel-resource-usage]\n" \
        "remark: vgl_sample.hip:1026:0:     VGPRs: 64 [-Rpass-analysis=kernel-resource-usage]\n" \
        "remark: vgl_sample.hip:1026:0:     VGPRs Spill: 20 [-Rpass-analysis=kernel-resource-usage]\n"
    b = "vgl_betachain.hip:80:1: remark: Function Name: _Zfoo [-Rpass-analysis=kernel-resource-usage]\n" \
        "vgl_betachain.hip:80:1: remark:     VGPRs: 64 [-Rpass-analysis=kernel-resource-usage]\n" \
        "vgl_betachain.hip:80:1: remark:     VGPRs Spill: 20 [-Rpass-analysis=kernel-resource-usage]\n"
    assert kernel_budget.parse(a) == kernel_budget.parse(b) == {"_Zfoo": {"vgprs": 64, "vgpr_spill": 20}}


def test_compare_reports_each_kind_of_regression():
    base = {"k": {"vgprs": 64, "sgpr_spill": 0, "vgpr_spill": 0, "scratch": 0, "lds": 100, "occupancy": 8, "agprs": 0}}
    assert kernel_budget.compare(base, base) == []
    for field, worse in (("vgprs", 65), ("sgpr_spill", 51), ("vgpr_spill", 1), ("scratch", 4), ("lds", 104), ("occupancy", 7)):
        cur = {"k": dict(base["k"], **{field: worse})}
        bad = kernel_budget.compare(cur, base)
        assert len(bad) == 1 and field.split("_")[0] in bad[0], (field, bad)
    assert "new kernel" in kernel_budget.compare({"k": base["k"], "k2": base["k"]}, base)[0]
    assert "no longer built" in kernel_budget.compare({}, base)[0]
    better = {"k": dict(base["k"], vgprs=60, occupancy=8, lds=0)}
    assert kernel_budget.compare(better, base) == []          # an improvement passes (and is worth an --update)


def test_every_shipped_kernel_within_its_budget(current):
    budget = json.load(open(kernel_budget.BUDGET))["kernels"]
    assert len(current) >= 80 and all("vgprs" in v and "occupancy" in v for v in current.values())
    bad = kernel_budget.compare(current, budget)
    assert not bad, "kernel resource budget exceeded (python tools/kernel_budget.py --update if intended):\n" + "\n".join(bad)


def test_builds_the_round_depends_on(current):
    """the numbers DESIGN.md quotes for the bench configurations' kernels, as hard limits"""
    want = {
        "k_sample_seg<1, 1, 2>": dict(vgprs=64, occupancy=8, scratch=0, vgpr_spill=0),     # C3 / C4 / gl1q: eight wavefronts per SIMD, nothing in scratch (round 6)
        "k_sample_seg<1, 1, 4>": dict(vgprs=64, occupancy=8, scratch=20),                 # qsi16 / alltags
        "k_sample<2, false, 1, false, 2>": dict(vgprs=64, occupancy=8),                   # the same build with the segment loop inside (pools that do not hold a wavefront's reads)
        "k_gl<5, 2, false, 8, 4>": dict(sgpr_spill=0, vgpr_spill=0, scratch=0),           # C5: the plain fused build spills nothing
        "k_gl<5, 2, false, 4, 4>": dict(sgpr_spill=0, vgpr_spill=0, scratch=0),
        "k_gl2<5>": dict(vgprs=64, scratch=0, occupancy=8),                               # fixed-q / C3 / C4 likelihoods
        "k_sample<0, false, 1, false, 1>": dict(scratch=0, occupancy=8),                  # fixed-q sampler
    }
    for name, lim in want.items():
        assert name in current, name
        for f, v in lim.items():
            if f == "occupancy":
                assert current[name][f] >= v, (name, f, current[name][f])
            else:
                assert current[name][f] <= v, (name, f, current[name][f])
