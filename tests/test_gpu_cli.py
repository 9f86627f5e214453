"""End-to-end runs of the C++ front end (vcfgl_amd/bin/vcfgl_hip: the reference's flags, VCF text in,
VCF text out, GPU simulation through the C ABI) in serial RNG mode against the reference's golden
VCFs, the way test/runTests.sh does it: `diff -I '^##'`.  Records must match as TEXT (this pins the
htslib-style float formatting), I16 tail-distance fields included."""
import gzip
import os
import subprocess

import pytest

import golden_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vcfgl_amd", "bin", "vcfgl_hip")
CASES = sorted(gu.REF_TESTS, key=lambda s: int(s[4:]))


def _mask_i16(line):
    f = line.split("\t")
    if len(f) > 7 and "I16=" in f[7]:
        parts = f[7].split(";")
        for k, p in enumerate(parts):
            if p.startswith("I16="):
                v = p[4:].split(",")
                parts[k] = "I16=" + ",".join(v[:12] + ["*"] * 4)
        f[7] = ";".join(parts)
    return "\t".join(f)


@pytest.mark.parametrize("name", CASES)
def test_cli_diff_against_reference_golden(name, tmp_path):
    t = gu.REF_TESTS[name]
    data = os.path.join(gu.REFVCF, "data")
    argv = []
    toks = t["args"].split()
    for i in range(0, len(toks), 2):
        flag, val = toks[i], toks[i + 1]
        if flag in ("--depths-file", "--qs-bins"):
            val = os.path.join(data, os.path.basename(val))
        argv += [flag, val]
    out = str(tmp_path / name)
    cmd = [BIN, "-i", os.path.join(data, t["input"]), "-o", out, "--rng-mode", "1"] + argv
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ours = [l.rstrip("\n") for l in open(out + ".vcf") if not l.startswith("##")]
    gold = [l.rstrip("\n") for l in open(os.path.join(gu.REFVCF, "reference", name, name + ".vcf")) if not l.startswith("##")]
    assert len(ours) == len(gold)
    for a, b in zip(ours, gold):
        assert a == b                                   # full text equality, I16 tail-distance fields included
    truth = os.path.join("/nonexistent")
    tgold = os.path.join(gu.REFVCF, "reference", name, name + ".truth.vcf")
    if "-printTruth 1" in t["args"] and os.path.exists(tgold):
        a = [l for l in open(out + ".truth.vcf") if not l.startswith("##")]
        b = [l for l in open(tgold) if not l.startswith("##")]
        assert a == b
    if t.get("pileup"):
        a = gzip.open(out + ".pileup.gz", "rt").read()
        b = gzip.open(os.path.join(gu.REFVCF, "reference", name, name + ".pileup.gz"), "rt").read()
        assert a == b


class _FloatText:
    """float32 bit pattern -> VCF text through the host program's own kputd()-style formatter (which
    tests/test_cli_format_cpu.py pins against every float token of the reference's golden files)."""

    def __init__(self):
        self.cache = {}

    def prime(self, bits):
        todo = sorted(set(bits) - set(self.cache))
        for i in range(0, len(todo), 400):
            r = subprocess.run([BIN, "--format-floats"] + ["%08x" % b for b in todo[i:i + 400]], capture_output=True, text=True, check=True)
            self.cache.update(zip(todo[i:i + 400], r.stdout.split("\n")))

    def __call__(self, bits):
        return self.cache[bits]


@pytest.mark.parametrize("mode", ["b", "u"])
@pytest.mark.parametrize("name", CASES)
def test_cli_bcf_output_decodes_to_the_reference_golden(name, mode, tmp_path):
    """--output-mode b / u (the reference's default is b): the BCF file, decoded by the specification-based
    reader of tests/bcf_reader.py, must give the golden VCF's records (gVCF blocks, truth file included)."""
    import bcf_reader
    t = gu.REF_TESTS[name]
    data = os.path.join(gu.REFVCF, "data")
    argv, toks = [], t["args"].split()
    for i in range(0, len(toks), 2):
        flag, val = toks[i], toks[i + 1]
        if flag in ("--depths-file", "--qs-bins"):
            val = os.path.join(data, os.path.basename(val))
        if flag in ("--output-mode", "-O"):
            val = mode
        argv += [flag, val]
    out = str(tmp_path / name)
    r = subprocess.run([BIN, "-i", os.path.join(data, t["input"]), "-o", out, "--rng-mode", "1"] + argv, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ft = _FloatText()
    for suffix in (".bcf", ".truth.bcf"):
        gold_path = os.path.join(gu.REFVCF, "reference", name, name + suffix.replace("bcf", "vcf"))
        if suffix == ".truth.bcf" and not ("-printTruth 1" in t["args"] and os.path.exists(gold_path)):
            continue
        rd = bcf_reader.Reader(out + suffix)
        assert rd.compressed == (mode == "b")
        bits = []
        for rec in rd.records():
            bits += [x for _, ty, v in rec["info"] if ty == 5 for x in v]
            bits += [x for _, ty, per in rec["fmt"] if ty == 5 for v in per for x in v]
        ft.prime(bits)
        rd = bcf_reader.Reader(out + suffix)
        ours = list(rd.vcf_lines(ft))
        gold = [l.rstrip("\n") for l in open(gold_path) if not l.startswith("#")]
        assert ours == gold


def test_cli_errors_like_the_reference(tmp_path):
    data = os.path.join(gu.REFVCF, "data")
    r = subprocess.run([BIN, "-i", os.path.join(data, "data2.vcf"), "-d", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "Error rate is not specified" in r.stderr
    r = subprocess.run([BIN, "-i", os.path.join(data, "data2.vcf"), "-d", "2", "-e", "0.1", "--gl-model", "1", "--precise-gl", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "not supported with genotype likelihood model 1" in r.stderr


def test_cli_tile_mode_independent_of_tile_size(tmp_path):
    data = os.path.join(gu.REFVCF, "data")
    outs = []
    for ts in (1, 3, 4096):
        out = str(tmp_path / f"t{ts}")
        r = subprocess.run([BIN, "-i", os.path.join(data, "data3.vcf"), "-o", out, "--seed", "42", "-d", "5", "-e", "0.01", "-explode", "1",
                            "-addPL", "1", "-addFormatAD", "1", "-O", "v", "--tile-sites", str(ts)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in open(out + ".vcf") if not l.startswith("##")])
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 5


def test_cli_threads_do_not_change_the_file(tmp_path):
    """--threads N encodes records and compresses BGZF blocks on N threads; the record bytes are the same
    (the header differs in its command line)"""
    import bcf_reader
    data = os.path.join(gu.REFVCF, "data")
    blobs = []
    for th in (1, 7):
        out = str(tmp_path / f"th{th}")
        r = subprocess.run([BIN, "-i", os.path.join(data, "data3.vcf"), "-o", out, "--seed", "42", "-d", "5", "-e", "0.01",
                            "-explode", "1", "-addPL", "1", "-addGP", "1", "-addFormatAD", "1", "-addQS", "1", "-O", "b", "--threads", str(th),
                            "--tile-sites", "5"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        rd = bcf_reader.Reader(out + ".bcf")
        assert rd.compressed
        blobs.append(rd.raw[rd.off:])
    assert blobs[0] == blobs[1] and len(blobs[0]) > 500


DOC = os.path.join(ROOT, "tests", "golden", "doc_error_qs")


@pytest.mark.parametrize("eq", [0, 1, 2])
def test_cli_tsv_dumps_against_documented_outputs(eq, tmp_path):
    """-printBasePickError / -printQsError / -printGlError / -printQScores (vcfgl.cpp:430-435, 533-554, 1745-1755;
    io.cpp:1089-1100): the three runs the reference documents in doc/error_qs.MD, their TSV listings (sorted
    there), pileups and VCF records reproduced as text through the GPU in serial RNG mode."""
    out = str(tmp_path / f"error_qs{eq}")
    cmd = [BIN, "-i", os.path.join(gu.REFVCF, "data", "data2.vcf"), "-o", out, "--rng-mode", "1", "--depth", "2",
           "--error-rate", "0.4", "--error-qs", str(eq), "-addFormatAD", "1", "-printPileup", "1", "-s", "42", "-O", "v",
           "-printBasePickError", "1", "-printQsError", "1", "-printGlError", "1", "-printQScores", "1"]
    if eq:
        cmd += ["--beta-variance", "1e-1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    want = sorted(open(os.path.join(DOC, f"details_qs{eq}.tsv")).read().splitlines())
    assert sorted(r.stdout.splitlines()) == want
    assert gzip.open(out + ".pileup.gz", "rt").read() == open(os.path.join(DOC, f"error_qs{eq}.pileup")).read()
    ours = [l.rstrip("\n") for l in open(out + ".vcf") if not l.startswith("##")]
    gold = [l.rstrip("\n") for l in open(os.path.join(DOC, f"error_qs{eq}.vcf")) if not l.startswith("##")]
    assert ours == gold


def test_cli_tsv_line_order_and_adjusted_scores(tmp_path):
    """Unsorted stdout: per site and sample, the lines of a read follow each other in the order qs_error_prob, qs,
    gl_error_prob (vcfgl.cpp:533-554); --adjust-qs 8 / 16 / 4 switch the printed score, the GL error probability
    and the pileup to the adjusted quality score."""
    base = [BIN, "-i", os.path.join(gu.REFVCF, "data", "data2.vcf"), "--rng-mode", "1", "--depth", "3", "--error-rate", "0.1",
            "--error-qs", "2", "--beta-variance", "1e-3", "-s", "7", "-O", "v", "-printPileup", "1",
            "-printQsError", "1", "-printGlError", "1", "-printQScores", "1"]
    r0 = subprocess.run(base + ["-o", str(tmp_path / "a")], capture_output=True, text=True, timeout=300)
    r1 = subprocess.run(base + ["-o", str(tmp_path / "b"), "--adjust-qs", "28", "--adjust-by", "0.499"], capture_output=True, text=True, timeout=300)
    assert r0.returncode == 0 and r1.returncode == 0, r0.stderr[-1000:] + r1.stderr[-1000:]
    l0, l1 = r0.stdout.splitlines(), r1.stdout.splitlines()
    assert len(l0) == len(l1) and len(l0) % 3 == 0 and len(l0) > 0
    import math
    n_adj = 0
    for k in range(0, len(l0), 3):
        a, b, c = (x.split("\t") for x in l0[k:k + 3])
        assert (a[0], b[0], c[0]) == ("qs_error_prob", "qs", "gl_error_prob") and a[1:5] == b[1:5] == c[1:5]
        a1, b1, c1 = (x.split("\t") for x in l1[k:k + 3])
        assert a1 == a                                          # same deviates
        ep = float(a[5])
        if 1e-5 < ep < 0.99 and abs((-10 * math.log10(ep)) % 1 - 0.5) > 0.01:   # away from the %f rounding of the printed deviate
            q, aq = int(-10 * math.log10(ep)), int(-10 * math.log10(ep) + 0.499)
            assert int(b[5]) == min(q, 63) and int(b1[5]) == min(aq, 63)
            n_adj += aq != q
    assert n_adj > 0
    pa = gzip.open(str(tmp_path / "a") + ".pileup.gz", "rt").read()
    pb = gzip.open(str(tmp_path / "b") + ".pileup.gz", "rt").read()
    assert pa != pb and len(pa) == len(pb)


def test_cli_documented_msprime_run(tmp_path):
    """doc/with_msprime.MD: 3 samples, --depth 10 --error-rate 0 --source 0 --seed 42; the listed records as text"""
    d = os.path.join(ROOT, "tests", "golden", "doc_msprime")
    out = str(tmp_path / "sim_source0")
    r = subprocess.run([BIN, "-i", os.path.join(d, "msprime_output.vcf"), "-O", "v", "-o", out, "--rng-mode", "1", "--depth", "10",
                        "--error-rate", "0", "--source", "0", "--seed", "42"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ours = [l.rstrip("\n") for l in open(out + ".vcf") if not l.startswith("##")]
    gold = [l.rstrip("\n") for l in open(os.path.join(d, "sim_source0.vcf")) if not l.startswith("##")]
    assert ours == gold


@pytest.mark.parametrize("mode", [0, 1])
def test_baseline_config_c1_cli_equals_oracle(oracle, mode, tmp_path):
    """BASELINE.json configs[0]: `test/data/data2.vcf --depth 4 --error-rate 0.01 -GL 1 --seed 42` -- the reference's own
    CPU-runnable case.  No golden file of the reference holds this run, so the host program's VCF text is compared with
    the oracle driven through the Python mirror of the record loop, in both RNG modes (in serial mode the oracle replays
    the reference's stream order, which the 14 golden VCFs pin)."""
    import numpy as np
    from vcfgl_amd import VcfglArgs, _abi
    from vcfgl_amd.recordloop import iter_sites
    from vcfgl_amd.vcfio import read_vcf
    inp = os.path.join(gu.REFVCF, "data", "data2.vcf")
    flags = ["--depth", "4", "--error-rate", "0.01", "-GL", "1", "--seed", "42"]
    args = VcfglArgs.from_argv(flags).validate()
    args.rng_mode, args.beta_sampler = mode, (_abi.VGL_BETA_STD if mode == 1 else _abi.VGL_BETA_RAND48)
    out = str(tmp_path / f"c1_{mode}")
    r = subprocess.run([BIN, "-i", inp, "-o", out, "-O", "v", "--rng-mode", str(mode)] + flags, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    vcf = read_vcf(inp)
    sites = list(iter_sites(vcf, args))
    tile = oracle.Oracle(args, len(vcf.samples)).simulate(0, np.stack([s.gt for s in sites]),
                                                          fields=[f for f, _, _ in _abi.TILE_FIELDS if f not in ("qs", "i16")])
    got = read_vcf(out + ".vcf")
    assert len(got.records) == int((tile.numpy("site_status") >= 0).sum()) > 0
    errs = gu.compare_with_golden(args, sites, tile, got)
    assert not errs, errs[:10]
    assert int(tile.numpy("fmt_dp").max()) >= 4            # depth-4 reads really reach GL model 1 (n up to ~10)


@pytest.mark.parametrize("case", ["plain", "gvcf", "pileup"])
def test_devices_do_not_change_the_output(case, tmp_path):
    """--devices a,b,...: one context and one host thread per device, tiles dealt round robin and written in site order.
    Every value depends only on the absolute site index, so the files must be byte-identical to the one-device run --
    plain records, gVCF blocks that run across tiles simulated by different contexts, and the pileup.  (On a one-GPU box
    the contexts share the device: the host-side machinery is the same.)"""
    data = os.path.join(gu.REFVCF, "data")
    flags = {"plain": ["-i", os.path.join(data, "data3.vcf"), "--depth", "6", "--error-rate", "0.01", "--error-qs", "2", "--beta-variance", "1e-5",
                       "-explode", "1", "-doUnobserved", "2", "-addPL", "1", "-addGP", "1", "-addQS", "1", "-addFormatAD", "1", "-addInfoAD", "1", "-printTruth", "1"],
             "gvcf": ["-i", os.path.join(data, "data2.vcf"), "--depth", "4", "--error-rate", "0.001", "-explode", "1", "-doUnobserved", "2", "-addPL", "1",
                      "-doGVCF", "1", "--gvcf-dps", "1,3"],
             "pileup": ["-i", os.path.join(data, "data3.vcf"), "--depth", "3", "--error-rate", "0.02", "--error-qs", "2", "--beta-variance", "1e-4",
                        "-explode", "1", "-printPileup", "1", "-printQScores", "1"]}[case]
    outs = {}
    for name, dev in (("one", ["--device", "0"]), ("two", ["--devices", "0,0"]), ("three", ["--devices", "0,0,0"])):
        out = str(tmp_path / name)
        r = subprocess.run([BIN, "-o", out, "-O", "v", "--seed", "42", "--rng-mode", "0", "--tile-sites", "3"] + dev + flags, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        body = [l for l in open(out + ".vcf") if not l.startswith("##")]
        extra = r.stdout
        if case == "pileup":
            extra += gzip.open(out + ".pileup.gz", "rt").read()
        if case == "plain":
            extra += "".join(l for l in open(out + ".truth.vcf") if not l.startswith("##"))
        outs[name] = (body, extra)
    assert len(outs["one"][0]) > 5
    assert outs["one"] == outs["two"] == outs["three"]
    r = subprocess.run([BIN, "-o", str(tmp_path / "x"), "-O", "v", "--seed", "42", "--rng-mode", "1", "--devices", "0,0"] + flags, capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "does not shard" in r.stderr                     # the serial draw order is one stream
