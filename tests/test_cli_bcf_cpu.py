"""--output-mode z / u / b of the host program on the GPU-free --depth inf path: the bgzip'd VCF
decompresses to the -O v text, and the BCF files, decoded by a reader written from the BCF2 / BGZF
specifications (tests/bcf_reader.py), give back the same records, truth file included."""
import gzip
import os
import struct
import subprocess

import pytest

import bcf_reader
import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vcfgl_amd", "bin", "vcfgl_hip")
DATA = os.path.join(gu.REFVCF, "data")
pytestmark = pytest.mark.skipif(not os.path.exists(BIN), reason="vcfgl_hip not built")

ARGV = ("--seed 42 --depth inf --error-rate 0 -explode 1 -doUnobserved 1 -printTruth 1 -addGP 1 -addPL 1 "
        "-addI16 0 -addQS 0 -addFormatDP 0").split()


def simple_float(bits):
    f = struct.unpack("<f", struct.pack("<I", bits))[0]
    return {0.0: "0", 1.0: "1", float("-inf"): "-inf"}[f]


def run(tmp_path, name, mode, vcf="data3.vcf", extra=()):
    out = str(tmp_path / name)
    r = subprocess.run([BIN, "-i", os.path.join(DATA, vcf), "-o", out, "-O", mode] + ARGV + list(extra), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    return out


def body(lines):
    return [l.rstrip("\n") for l in lines if not l.startswith("#")]


@pytest.mark.parametrize("vcf", ["data3.vcf", "data1.vcf"])
def test_output_modes_agree(tmp_path, vcf):
    v = run(tmp_path, "v", "v", vcf)
    want = body(open(v + ".vcf"))
    want_truth = body(open(v + ".truth.vcf"))
    assert len(want) >= 5

    z = run(tmp_path, "z", "z", vcf)
    raw = open(z + ".vcf.gz", "rb").read()
    assert body(b"".join(bcf_reader.bgzf_blocks(raw)).decode().splitlines()) == want
    assert body(gzip.open(z + ".truth.vcf.gz", "rt")) == want_truth

    for mode in ("u", "b"):
        o = run(tmp_path, mode, mode, vcf)
        rd = bcf_reader.Reader(o + ".bcf")
        assert rd.compressed == (mode == "b")
        assert list(rd.vcf_lines(simple_float)) == want
        assert rd.header[-1].split("\t")[9:] == open(v + ".vcf").read().split("#CHROM")[1].split("\n")[0].split("\t")[9:]
        # every dictionary line carries its index; PASS is entry 0
        assert rd.dict[0] == "PASS"
        for h in rd.header:
            if h.startswith(("##FILTER=", "##INFO=", "##FORMAT=", "##contig=")):
                assert ",IDX=" in h, h
        rt = bcf_reader.Reader(o + ".truth.bcf")
        assert list(rt.vcf_lines(simple_float)) == want_truth


def test_default_output_mode_is_compressed_bcf(tmp_path):
    out = str(tmp_path / "dflt")                          # io.cpp:776-777: --output-mode defaults to b
    r = subprocess.run([BIN, "-i", os.path.join(DATA, "data3.vcf"), "-o", out] + ARGV, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    assert bcf_reader.Reader(out + ".bcf").compressed


def test_record_layout_fields(tmp_path):
    """Fixed-width fields of the records against the -O v text: 0-based POS, rlen = len(REF), missing QUAL,
    typed FORMAT arrays; genotypes of the truth file as (allele+1)<<1|phased."""
    v = run(tmp_path, "v", "v")
    o = run(tmp_path, "u", "u")
    rd = bcf_reader.Reader(o + ".bcf")
    txt = [l.split("\t") for l in body(open(v + ".vcf"))]
    for rec, src in zip(rd.records(), txt):
        assert rec["pos0"] == int(src[1]) - 1 and rec["rlen"] == len(src[3]) and rec["qual"] == bcf_reader.F_MISSING
        assert [k for k, _, _ in rec["fmt"]] == ["GL", "GP", "PL"]
        assert all(t == 5 for k, t, _ in rec["fmt"] if k in ("GL", "GP"))
        assert [t for k, t, _ in rec["fmt"] if k == "PL"] == [2]                       # 255 needs int16
    rt = bcf_reader.Reader(o + ".truth.bcf")
    for rec, src in zip(rt.records(), [l.split("\t") for l in body(open(v + ".truth.vcf"))]):
        (k, t, per), = rec["fmt"]
        assert k == "GT" and t == 1
        for s, g in enumerate(src[9:]):
            a, b = g.split("|")
            assert per[s] == [(int(a) + 1) << 1, ((int(b) + 1) << 1) | 1]


def test_multithreading_is_refused_for_text_output(tmp_path):
    r = subprocess.run([BIN, "-i", os.path.join(DATA, "data3.vcf"), "-o", str(tmp_path / "x"), "-O", "v", "--threads", "4"] + ARGV,
                       capture_output=True, text=True)
    assert r.returncode != 0 and "Multithreading is not supported for VCF output" in r.stderr       # io.cpp:1206-1210


@pytest.mark.parametrize("vals,bt", [
    ([0, 127, -120], 1), ([128, 0], 2), ([-121, 5], 2), ([32767, -32760], 2), ([32768], 3), ([-32761, 1], 3),
    ([2147483647, -2147483640], 3), ([".", 7, "."], 1), ([3] * 15, 1), ([1] * 14, 1), (list(range(-100, 100)), 1), ([300] * 130, 2),
])
def test_integer_type_choice_at_the_boundaries(tmp_path, vals, bt):
    """htslib's bcf_enc_vint: int8 holds [-120, 127], int16 [-32760, 32767] (the low values are reserved for
    missing / end-of-vector); vectors of 15 or more elements carry their length as a typed integer."""
    for mode in ("u", "b"):
        out = str(tmp_path / f"st_{mode}.bcf")
        r = subprocess.run([BIN, "--encode-selftest", mode, out] + [str(v) for v in vals], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        rd = bcf_reader.Reader(out)
        (rec,) = list(rd.records())
        want = [None if v == "." else v for v in vals]
        (k, t, per), = rec["fmt"]
        assert k == "X" and t == bt and per == [want, want[::-1]]
        (ik, it, iv), = rec["info"]
        assert ik == "Y" and iv == want and (it == bt or len(vals) == 1)
        assert rec["alleles"] == ["A", "C", "<*>"] and rec["filter"] == ["PASS"] and rec["id"] == "rs1" and rec["pos0"] == 4


@pytest.mark.parametrize("vcf,src", [("data3.vcf", 0), ("data2.vcf", 0), ("data5_acgt_multiallelic.vcf", 1), ("data1.vcf", 0)])
def test_bcf_input_round_trip(tmp_path, vcf, src):
    """BCF (raw and BGZF) as INPUT: the truth file of a run holds the decoded input records; fed back as
    BCF (its alleles are already A/C/G/T, hence --source 1) it gives the records of the VCF-text run."""
    base = ["--seed", "1", "--depth", "inf", "-e", "0", "-printTruth", "1", "-doUnobserved", "1"]
    ref = str(tmp_path / "ref")
    r = subprocess.run([BIN, "-i", os.path.join(DATA, vcf), "-o", ref, "-O", "v", "--source", str(src)] + base, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    for mode in ("u", "b"):
        first = str(tmp_path / ("first_" + mode))
        r = subprocess.run([BIN, "-i", os.path.join(DATA, vcf), "-o", first, "-O", mode, "--source", str(src)] + base, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        again = str(tmp_path / ("again_" + mode))
        r = subprocess.run([BIN, "-i", first + ".truth.bcf", "-o", again, "-O", "v", "--source", "1"] + base, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        assert body(open(again + ".vcf")) == body(open(ref + ".vcf"))
        assert body(open(again + ".truth.vcf")) == body(open(ref + ".truth.vcf"))


def test_run_log_is_written(tmp_path):
    out = run(tmp_path, "log", "v")                         # io.cpp:1031: <prefix>.arg
    txt = open(out + ".arg").read()
    assert "Command: vcfgl_hip" in txt and "Simulation finished successfully" in txt and out + ".vcf" in txt and "truth" in txt


def test_bcf_in_bcf_out_with_an_info_key_defined_after_format_gt(tmp_path):
    """A BCF input keeps its header lines, IDX= attributes included; the output drops FORMAT/GT, which shifts the index of
    every key defined after it.  The indices written must be the ones the records are encoded with."""
    src = tmp_path / "in.vcf"
    src.write_text("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n##contig=<ID=c1,length=9>\n"
                   "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
                   "##INFO=<ID=AC,Number=A,Type=Integer,Description=\"after GT\">\n##INFO=<ID=ZZ,Number=1,Type=Float,Description=\"after GT too\">\n"
                   "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\ts2\n"
                   "c1\t2\t.\tA\tC\t.\tPASS\tAC=3;ZZ=0.5\tGT\t0|1\t1|1\n"
                   "c1\t5\trs9\tG\tT\t.\tPASS\tAC=1\tGT\t0|0\t0|1\n")
    base = ["--seed", "1", "--depth", "inf", "-e", "0", "-printTruth", "1", "-doUnobserved", "1", "--source", "1"]
    first = str(tmp_path / "first")
    r = subprocess.run([BIN, "-i", str(src), "-o", first, "-O", "u"] + base, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    hdr_in = bcf_reader.Reader(first + ".truth.bcf").header
    gt_idx = [int(h.split(",IDX=")[1].rstrip(">")) for h in hdr_in if h.startswith("##FORMAT=<ID=GT,")][0]
    ac_idx = [int(h.split(",IDX=")[1].rstrip(">")) for h in hdr_in if h.startswith("##INFO=<ID=AC,")][0]
    assert gt_idx < ac_idx                                        # the situation of the finding: GT's index is below AC's
    text = str(tmp_path / "text")
    r = subprocess.run([BIN, "-i", first + ".truth.bcf", "-o", text, "-O", "v"] + base, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    for mode in ("u", "b"):
        again = str(tmp_path / ("again_" + mode))
        r = subprocess.run([BIN, "-i", first + ".truth.bcf", "-o", again, "-O", mode] + base, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1500:]
        rd = bcf_reader.Reader(again + ".bcf")
        idx = {}
        for h in rd.header:
            if h.startswith(("##FILTER=", "##INFO=", "##FORMAT=")):
                assert h.count(",IDX=") == 1, h
                i = int(h.split(",IDX=")[1].rstrip(">"))
                assert i not in idx, (h, idx[i])                  # no two dictionary lines share an index
                idx[i] = h
        recs = list(bcf_reader.Reader(again + ".bcf").records())
        assert [[k for k, _, _ in r_["info"]] for r_ in recs] == [["AC", "ZZ"], ["AC"]]
        assert [r_["info"][0][2] for r_ in recs] == [[3], [1]]
        assert [r_["id"] for r_ in recs] == [".", "rs9"]
        assert list(bcf_reader.Reader(again + ".bcf").vcf_lines(lambda b: {0.0: "0", 1.0: "1", float("-inf"): "-inf", 0.5: "0.5"}[struct.unpack("<f", struct.pack("<I", b))[0]])) == body(open(text + ".vcf"))


def test_missing_id_is_a_zero_length_string(tmp_path):
    """htslib encodes ID '.' as a typed string of length 0 (byte 0x07), not as the one-character string '.'"""
    o = run(tmp_path, "id", "u")
    raw = open(o + ".bcf", "rb").read()
    l_text = struct.unpack_from("<I", raw, 5)[0]
    off = 9 + l_text
    assert raw[off + 8 + 24] == 0x07                           # l_shared, l_indiv, six fixed fields, then the ID


def test_explode_refuses_unsorted_or_duplicate_positions(tmp_path):
    """-explode 1 on a record that is not after its predecessor: the reference never leaves `while (n_in != pos)`
    (vcfgl.cpp:1481); here it is an error message instead of an allocation until the machine runs out of memory."""
    src = tmp_path / "dup.vcf"
    src.write_text("##fileformat=VCFv4.2\n##contig=<ID=c1,length=9>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
                   "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\ts1\n"
                   "c1\t3\t.\t0\t1\t.\tPASS\t.\tGT\t0|1\nc1\t3\t.\t0\t1\t.\tPASS\t.\tGT\t1|1\n")
    r = subprocess.run([BIN, "-i", str(src), "-o", str(tmp_path / "o"), "-O", "v", "--seed", "1", "--depth", "inf", "-e", "0", "-explode", "1"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot be exploded" in r.stderr
