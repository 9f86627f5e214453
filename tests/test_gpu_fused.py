"""The fused build of k_gl (GL model 2, one fixed quality score, default tag surface, every mean depth below 12, 128 < N <= 512): one
workgroup (256 or 512 threads) samples a site's reads, orders its alleles and evaluates the likelihoods without staging anything in HBM.  Same oracle
parity as the three-kernel path (bit-exact integers and GL), which `VGL_NO_FUSE=1` still runs."""
import os

import numpy as np
import pytest

import synth
from vcfgl_amd import Simulator, VcfglArgs, _abi
from test_gpu_parity import assert_parity, run_both

pytestmark = pytest.mark.gpu

TAGS = dict(add_gp=1, add_pl=1, add_info_dp=1, add_fmt_ad=1, add_info_ad=1)       # (no QS / I16 / strand tags: those take the three-kernel path)


def _is_fused(args, N, gt, hooks=False):
    """vgl_ctx_info() says which build a context launches (ABI 5); the timing buckets agree: nothing runs in k_sample's or k_site's"""
    import torch
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    sim = Simulator(args, N, device=0, max_sites_per_tile=gt.shape[0], hooks=hooks)
    info = sim.info()
    sim.timing(True)
    g = torch.from_numpy(np.ascontiguousarray(gt)).to("cuda:0")
    tile = sim.new_tile(gt.shape[0], fields=["fmt_dp", "gl"], device="cuda:0")
    for _ in range(3):
        sim.simulate_device(0, g, tile); sim.check()
    ms, n = sim.kernel_ms(reset=True)
    sim.close()
    assert n[_abi.T_GL] == 3
    if info["fused"]:
        assert info["sample_lean"] >= 1 and info["fused_split"] >= 1
        assert max(ms[_abi.T_SAMPLE], ms[_abi.T_SITE], ms[_abi.T_REDO]) / 3 < 0.03               # empty buckets: two events back to back (~5 us), no kernel
    return bool(info["fused"])


@pytest.mark.parametrize("N", [128, 129, 200, 256, 257, 300, 500, 511, 512, 513, 777, 1000, 1024, 1025, 1537, 2000, 4096, 4097])
@pytest.mark.parametrize("depth", [0.3, 5.0, 11.9, 12.0, 20.0, 45.0, 70.0])
def test_fused_shapes_and_depths(oracle, N, depth, monkeypatch):
    """(round 4) sites of more than 512 samples are split over consecutive workgroups (513 ... 2048: 2 ... 4 of them; beyond -- 4096, 4097 --
    the three kernels run: at eight workgroups per site the fused kernel measured no faster), depths of 12 and more come from k_depth,
    up to 128 staged reads (depth 45: 115; depth 70 runs the three kernels)"""
    if N > 1100 and depth not in (5.0, 20.0):
        pytest.skip("wide sites: two depths are enough")
    args = VcfglArgs(seed=11, depth=depth, error_rate=0.01, **TAGS)
    gt = synth.binary_sites(5, 24 if N > 1100 else 60, N)
    if depth >= 12.0:
        monkeypatch.setenv("VGL_FUSE_DEEP", "1")                  # depth 12 and more runs fused only behind this hook (three kernels are faster there)
    want, got = run_both(oracle, args, gt, site0=5, hooks=depth >= 12.0)
    assert_parity(want, got)


def test_the_fused_kernel_is_the_one_that_runs():
    gt = np.zeros((8192, 500), dtype=np.uint8)
    assert _is_fused(VcfglArgs(seed=11, depth=5.0, error_rate=0.01, **TAGS), 500, gt)
    assert not _is_fused(VcfglArgs(seed=11, depth=5.0, error_rate=0.01, add_qs=1, **TAGS), 500, gt)          # -addQS needs the per-base quality sums
    assert _is_fused(VcfglArgs(seed=11, depth=5.0, error_rate=0.01, **TAGS), 200, np.zeros((8192, 200), dtype=np.uint8))             # 256 threads per site
    assert _is_fused(VcfglArgs(seed=11, depth=5.0, error_rate=0.01, **TAGS), 600, np.zeros((2048, 600), dtype=np.uint8))             # (round 4) two workgroups per site
    assert not _is_fused(VcfglArgs(seed=11, depth=20.0, error_rate=0.01, **TAGS), 1000, np.zeros((2048, 1000), dtype=np.uint8))      # depth 12 and more: the three kernels measured faster (tools/fuse_ab.sh)
    os.environ["VGL_FUSE_DEEP"] = "1"                             # (the deep fused build stays, behind a hook of the -DVGL_TEST_HOOKS library)
    try:
        assert _is_fused(VcfglArgs(seed=11, depth=20.0, error_rate=0.01, **TAGS), 1000, np.zeros((2048, 1000), dtype=np.uint8), hooks=True)    # k_depth's draws, 72 staged reads
        assert not _is_fused(VcfglArgs(seed=11, depth=70.0, error_rate=0.01, **TAGS), 500, np.zeros((512, 500), dtype=np.uint8), hooks=True)   # 153 staged reads: three kernels
    finally:
        del os.environ["VGL_FUSE_DEEP"]
    assert _is_fused(VcfglArgs(seed=11, depth=5.0, error_rate=0.01, **TAGS), 2048, np.zeros((512, 2048), dtype=np.uint8))            # four workgroups per site
    assert not _is_fused(VcfglArgs(seed=11, depth=5.0, error_rate=0.01, **TAGS), 2049, np.zeros((512, 2049), dtype=np.uint8))        # five: three kernels (measured no faster from eight on)
    assert not _is_fused(VcfglArgs(seed=11, depth=5.0, error_rate=0.01, error_qs=1, beta_variance=1e-5, **TAGS), 500, gt)            # a per-site draw: three kernels


@pytest.mark.parametrize("alone", ["255", "5", "2"])            # every part / parts 0 and 2 / part 1 only count as neighbours that never arrive
@pytest.mark.parametrize("N,depth", [(600, 5.0), (1000, 20.0), (1500, 14.0), (2048, 3.0)])
def test_split_fused_workgroup_that_does_not_wait(oracle, N, depth, alone, monkeypatch):
    """A site split over several fused workgroups: each adds its per-base depth sums to the site's record and waits, bounded, for the
    others; one that gives up samples the others' depths itself (nothing may depend on two workgroups being resident together).
    VGL_DEBUG_FUSE_ALONE=mask (hooks build) makes the workgroups treat the parts in the mask as absent: all of them, or a subset, so that
    a workgroup adds the published sums of some neighbours and recomputes the others -- same bits."""
    monkeypatch.setenv("VGL_DEBUG_FUSE_ALONE", alone)
    monkeypatch.setenv("VGL_FUSE_DEEP", "1")
    args = VcfglArgs(seed=23, depth=depth, error_rate=0.02, **TAGS)
    want, got = run_both(oracle, args, synth.acgt_sites(30, N, seed=N, missing=0.02), hooks=True)
    assert_parity(want, got)


@pytest.mark.parametrize("kw", [dict(do_unobserved=0), dict(do_unobserved=1), dict(do_unobserved=2), dict(do_unobserved=3), dict(do_unobserved=4),
                                dict(rm_invar_sites=4), dict(rm_empty_sites=1, depth=0.01), dict(error_rate=0.2), dict(error_rate=0.0), dict(error_qs=1, beta_variance=1e-5)])
def test_fused_site_options(oracle, kw):
    """allele-order options (4 or 5 alleles per site), skipped sites, many errors (three and four bases per evaluation), no errors;
    --error-qs 1 (a per-site draw) is NOT fused and must still be right"""
    a = dict(seed=5, depth=4.0, error_rate=0.01)
    a.update(TAGS); a.update(kw)
    want, got = run_both(oracle, VcfglArgs(**a), synth.acgt_range(100, 80, 400, missing=0.03), site0=100)
    assert_parity(want, got)


def test_fused_per_sample_depths_and_missing_genotypes(oracle):
    N = 333
    rng = np.random.default_rng(3)
    depths = list(rng.uniform(0.0, 11.5, N))
    args = VcfglArgs(seed=9, depths=depths, error_rate=0.02, **TAGS)
    want, got = run_both(oracle, args, synth.acgt_range(0, 50, N, missing=0.2))
    assert_parity(want, got)


@pytest.mark.parametrize("N", [500, 300, 130])                             # 300, 130: wavefronts of the site's workgroup beyond its samples
@pytest.mark.parametrize("layout", [_abi.VGL_LAYOUT_PLANES, _abi.VGL_LAYOUT_SAMPLE_MAJOR])
def test_fused_layouts_and_narrow_pl(oracle, layout, N):
    S = 70
    args = VcfglArgs(seed=21, depth=5.0, error_rate=0.01, do_unobserved=2, add_pl=1, add_fmt_ad=1, out_layout=layout)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    gt = np.zeros((S, N), dtype=np.uint8)                                  # config C5's hom-ref sites
    fields = ["site_status", "n_alleles", "alleles2acgt", "fmt_dp", "gl", "pl", "pl_u8", "fmt_ad"]
    o = oracle.Oracle(args, N)
    want = o.simulate(0, gt, fields=fields)
    sim = Simulator(args, N, device=0, max_sites_per_tile=S)
    got = sim.simulate(0, gt, fields=fields)
    sim.close()
    MULTI = {"gl": "G", "pl": "G", "pl_u8": "G", "fmt_ad": "A"}
    st, na = want.numpy("site_status"), want.numpy("n_alleles")
    for f in fields:
        if layout == _abi.VGL_LAYOUT_SAMPLE_MAJOR and f in MULTI:             # the record arrays (what lies behind them in a slab is unspecified)
            for i in range(S):
                nA = int(na[i]) if st[i] >= 0 else 0
                nk = nA * (nA + 1) // 2 if MULTI[f] == "G" else nA
                assert np.array_equal(np.ascontiguousarray(want.site_records(f, i, nk)).view(np.uint8),
                                      np.ascontiguousarray(got.site_records(f, i, nk)).view(np.uint8)), (f, i)
        else:
            assert np.array_equal(want.numpy(f).view(np.uint8), got.numpy(f).view(np.uint8)), f


def test_fused_equals_three_kernel_path_and_tiling(oracle):
    """3000 sites x 500 samples at config C5's flags: the fused kernel, the three-kernel path (VGL_NO_FUSE=1) and a run in ragged tiles agree bit for bit"""
    N, S = 500, 3000
    args = VcfglArgs(seed=42, depth=5.0, error_rate=0.01, do_unobserved=2, add_pl=1, add_fmt_ad=1, add_info_ad=1)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    gt = synth.binary_sites(0, S, N)
    gt[::3] = 0
    fields = ["site_status", "n_alleles", "n_alleles_obs", "alleles2acgt", "info_dp", "info_ad", "fmt_dp", "gl", "pl", "fmt_ad"]
    sim = Simulator(args, N, device=0, max_sites_per_tile=S)
    a = sim.simulate(0, gt, fields=fields)
    sim.close()
    os.environ["VGL_NO_FUSE"] = "1"                           # (an override of the -DVGL_TEST_HOOKS build; the shipped library ignores it)
    try:
        sim = Simulator(args, N, device=0, max_sites_per_tile=S, hooks=True)
        assert sim.info()["fused"] == 0
        b = sim.simulate(0, gt, fields=fields)
        sim.close()
        sim = Simulator(args, N, device=0, max_sites_per_tile=S)
        assert sim.info()["fused"] == 1
        sim.close()
    finally:
        del os.environ["VGL_NO_FUSE"]
    for f in fields:
        assert np.array_equal(a.numpy(f).view(np.uint8), b.numpy(f).view(np.uint8)), f
    sim = Simulator(args, N, device=0, max_sites_per_tile=1100)
    s0 = 0
    for n in (1, 1100, 7, 900, 992):
        t = sim.simulate(s0, gt[s0:s0 + n], fields=fields)
        for f in fields:
            assert np.array_equal(t.numpy(f).view(np.uint8), a.numpy(f)[s0:s0 + n].view(np.uint8)), (f, s0)
        s0 += n
    sim.close()
    assert s0 == S


def test_fused_through_the_host_program(tmp_path):
    """config C5's shape through vcfgl_hip (vgl_simulate_tile_async, sample-major slabs, pl_u8, gVCF blocks): 300 samples, an exploded
    contig, --tile-sites below the site count -- the files written with the fused kernel and with VGL_NO_FUSE=1 are the same bytes"""
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    BIN = os.path.join(ROOT, "vcfgl_amd", "bin", "vcfgl_hip")
    N, L = 300, 3000
    rng = np.random.default_rng(4)
    inp = str(tmp_path / "in.vcf")
    with open(inp, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n")
        fh.write(f"##contig=<ID=chr1,length={L}>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n")
        fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"s{i}" for i in range(N)) + "\n")
        for p in sorted(rng.choice(np.arange(1, L), size=40, replace=False)):
            g = rng.integers(0, 2, size=(N, 2))
            fh.write(f"chr1\t{p}\t.\t0\t1\t.\tPASS\t.\tGT\t" + "\t".join(f"{a}|{b}" for a, b in g) + "\n")
    outs = []
    for nofuse in (False, True):
        env = dict(os.environ)
        if nofuse:                                            # the override exists in the -DVGL_TEST_HOOKS build: preloaded, its entry points take precedence
            env["VGL_NO_FUSE"] = "1"
            env["LD_PRELOAD"] = os.path.join(ROOT, "vcfgl_amd", "lib", "libvcfgl_hip_hooks.so")
        for flags, tag in ((["-explode", "1", "-doGVCF", "1", "--gvcf-dps", "5,10,20", "-doUnobserved", "2", "-addPL", "1"], "g"),
                           (["-explode", "1", "-doUnobserved", "2", "-addPL", "1", "-addGP", "1", "-addFormatAD", "1", "-addInfoAD", "1"], "p")):
            out = str(tmp_path / f"o_{tag}_{int(nofuse)}")
            r = subprocess.run([BIN, "-i", inp, "-o", out, "-O", "v", "--seed", "42", "--depth", "5", "--error-rate", "0.01", "--tile-sites", "700", "--verbose", "1"] + flags,
                               capture_output=True, text=True, timeout=600, env=env)
            assert r.returncode == 0, r.stderr[-800:]
            # the comparison below is only worth something if the preloaded hooks build really switched the fused kernel off
            dev_lines = [l for l in r.stderr.splitlines() if l.startswith("[device ")]
            assert dev_lines, r.stderr[-800:]
            if nofuse:
                assert all(", fused 0 " in l for l in dev_lines), dev_lines
            elif tag == "g":
                assert all(", fused 1 " in l for l in dev_lines), dev_lines
            body = [l for l in open(out + ".vcf") if not l.startswith("##")]
            outs.append((tag, nofuse, body))
    by = {(t, n): b for t, n, b in outs}
    for t in ("g", "p"):
        assert len(by[(t, False)]) > 100
        assert by[(t, False)] == by[(t, True)], t
