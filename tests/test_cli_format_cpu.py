"""CPU-side check of the C++ front end's VCF float formatter (htslib kputd semantics) against every
float token the reference's golden VCFs contain: format(float32(token)) must give the token back."""
import os
import struct
import subprocess

import numpy as np
import pytest

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vcfgl_amd", "bin", "vcfgl_hip")


def _tokens():
    toks = set()
    refdir = os.path.join(gu.REFVCF, "reference")
    for d in sorted(os.listdir(refdir)):
        path = os.path.join(refdir, d, d + ".vcf")
        if not os.path.exists(path):
            continue
        for line in open(path):
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            for kv in f[7].split(";"):
                k, _, v = kv.partition("=")
                if k in ("QS", "I16"):
                    toks.update(v.split(","))
            keys = f[8].split(":")
            for smp in f[9:]:
                for k, v in zip(keys, smp.split(":")):
                    if k in ("GL", "GP"):
                        toks.update(v.split(","))
    toks.discard(".")
    return sorted(toks)


@pytest.mark.skipif(not os.path.exists(BIN), reason="vcfgl_hip not built")
def test_float_formatter_round_trips_every_golden_token():
    toks = _tokens()
    assert len(toks) > 500
    hexes = [format(struct.unpack("<I", struct.pack("<f", np.float32(float(t))))[0], "08x") for t in toks]
    out = []
    for i in range(0, len(hexes), 400):
        r = subprocess.run([BIN, "--format-floats"] + hexes[i:i + 400], capture_output=True, text=True, check=True)
        out += r.stdout.split("\n")[:-1]
    bad = [(t, o) for t, o in zip(toks, out) if t != o]
    assert not bad, bad[:20]


@pytest.mark.skipif(not os.path.exists(BIN), reason="vcfgl_hip not built")
def test_depth_inf_matches_reference_golden(tmp_path):
    """--depth inf (no sampling: true genotype gets GL 0) is pure host formatting and needs no GPU:
    test4 of the reference's suite (runTests.sh:391-416), records compared as text"""
    data = os.path.join(gu.REFVCF, "data")
    out = str(tmp_path / "test4")
    argv = ("--seed 42 --output-mode v --depth inf --error-rate 0 --gl-model 1 --precise-gl 0 -explode 1 --rm-empty-sites 1 "
            "--adjust-qs 1 -doUnobserved 1 -printTruth 1 -addGP 1 -addPL 1 -addI16 0 -addQS 0 -addFormatDP 0").split()
    r = subprocess.run([BIN, "-i", os.path.join(data, "data3.vcf"), "-o", out] + argv, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    ours = [l for l in open(out + ".vcf") if not l.startswith("##")]
    gold = [l for l in open(os.path.join(gu.REFVCF, "reference", "test4", "test4.vcf")) if not l.startswith("##")]
    assert ours == gold
    ours = [l for l in open(out + ".truth.vcf") if not l.startswith("##")]
    gold = [l for l in open(os.path.join(gu.REFVCF, "reference", "test4", "test4.truth.vcf")) if not l.startswith("##")]
    assert ours == gold


@pytest.mark.skipif(not os.path.exists(BIN), reason="vcfgl_hip not built")
def test_flag_matching_follows_the_reference():
    """io.cpp:538-752: the long simulation flags are compared with strcasecmp, the short / common ones with strcmp; -v / --version
    and -vv print and exit 0 (no GPU needed for any of this)"""
    data = os.path.join(gu.REFVCF, "data", "data2.vcf")
    r = subprocess.run([BIN, "-i", data, "-d", "2", "-e", "0.1", "--GL-Model", "1", "--PRECISE-GL", "1"], capture_output=True, text=True)
    assert r.returncode == 1 and "not supported with genotype likelihood model 1" in r.stderr
    r = subprocess.run([BIN, "-I", data, "-d", "2", "-e", "0.1"], capture_output=True, text=True)        # -i is case-sensitive
    assert r.returncode == 1 and "Unknown argument" in r.stderr
    for flag in ("-v", "--version", "-vv"):
        r = subprocess.run([BIN, flag], capture_output=True, text=True)
        assert r.returncode == 0 and "ABI" in r.stderr
