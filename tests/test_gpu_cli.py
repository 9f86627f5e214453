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
