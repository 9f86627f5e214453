"""Host program under AddressSanitizer + UBSan on the CPU-only path (--depth inf needs no GPU):
the reference test-suite case, hand-made malformed VCFs and seeded random mutations of a valid VCF.
A malformed input must end in an error message and a non-zero exit code, never in a memory error.
(Sanitizers are not available on the GPU pool; the device path is covered by the -m gpu tests.)"""
import os
import random
import subprocess

import pytest

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vcfgl_amd", "csrc")
LIB = os.path.join(ROOT, "vcfgl_amd", "lib")
DATA = os.path.join(gu.REFVCF, "data")
SAN_MARKS = ("AddressSanitizer", "runtime error", "LeakSanitizer")


@pytest.fixture(scope="module")
def asan_bin(tmp_path_factory):
    if not os.path.exists(os.path.join(LIB, "libvcfgl_hip.so")):
        subprocess.check_call(["make", "-C", CSRC, "all"])
    out = str(tmp_path_factory.mktemp("asan") / "vcfgl_asan")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-I" + os.path.join(ROOT, "include"), "-o", out, os.path.join(CSRC, "host", "vcfgl_main.cpp"),
                           "-L" + LIB, "-lvcfgl_hip", "-lz", "-pthread", "-Wl,-rpath," + LIB])
    return out


def run(binary, vcf, out, extra=()):
    r = subprocess.run([binary, "-i", vcf, "-o", out, "--seed", "1", "--depth", "inf", "-e", "0", "-O", "v", *extra],
                       capture_output=True, text=True, errors="replace", timeout=60)
    assert not any(m in r.stderr or m in r.stdout for m in SAN_MARKS), r.stderr[-3000:]
    return r


def test_reference_case_under_sanitizers(asan_bin, tmp_path):
    out = str(tmp_path / "t4")
    argv = ("--output-mode v --gl-model 1 --precise-gl 0 -explode 1 --rm-empty-sites 1 --adjust-qs 1 -doUnobserved 1 "
            "-printTruth 1 -addGP 1 -addPL 1 -addI16 0 -addQS 0 -addFormatDP 0").split()
    r = subprocess.run([asan_bin, "-i", os.path.join(DATA, "data3.vcf"), "-o", out, "--seed", "42", "--depth", "inf",
                        "--error-rate", "0"] + argv, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and not any(m in r.stderr for m in SAN_MARKS), r.stderr[-3000:]
    ours = [l for l in open(out + ".vcf") if not l.startswith("##")]
    gold = [l for l in open(os.path.join(gu.REFVCF, "reference", "test4", "test4.vcf")) if not l.startswith("##")]
    assert ours == gold
    for mode in ("z", "u", "b"):                               # the bgzip / BCF writers under the sanitizers
        r = subprocess.run([asan_bin, "-i", os.path.join(DATA, "data3.vcf"), "-o", out + mode, "--seed", "42", "--depth", "inf",
                            "--error-rate", "0", "-O", mode] + argv[2:], capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and not any(m in r.stderr for m in SAN_MARKS), r.stderr[-3000:]


def test_malformed_inputs_fail_cleanly(asan_bin, tmp_path):
    src = open(os.path.join(DATA, "data3.vcf")).read()
    lines = src.split("\n")
    body = [i for i, l in enumerate(lines) if l and not l.startswith("#")]
    cases = {
        "empty": "",
        "no_chrom_line": "\n".join(l for l in lines if not l.startswith("#CHROM")),
        "allele_index_out_of_range": src.replace("0|1", "0|9", 1),
        "no_gt_format": src.replace("\tGT\t", "\tXX\t"),
        "fewer_sample_columns": "\n".join(l.rsplit("\t", 1)[0] if i in body else l for i, l in enumerate(lines)),
        "more_sample_columns": "\n".join(l + "\t0|0" if i in body else l for i, l in enumerate(lines)),
        "nine_columns": "\n".join("\t".join(l.split("\t")[:9]) if i in body else l for i, l in enumerate(lines)),
        "empty_ref": "\n".join("\t".join(l.split("\t")[:3] + [""] + l.split("\t")[4:]) if i in body else l for i, l in enumerate(lines)),
        "six_alleles": "\n".join("\t".join(l.split("\t")[:4] + ["1,2,3,4,5"] + l.split("\t")[5:]) if i in body else l for i, l in enumerate(lines)),
    }
    for name, text in cases.items():
        f = str(tmp_path / (name + ".vcf"))
        open(f, "w").write(text)
        r = run(asan_bin, f, str(tmp_path / ("out_" + name)))
        assert r.returncode != 0, name
        assert "ERROR" in (r.stderr + r.stdout), name
    # a file cut in the middle of a record: either the complete records are simulated or an error is reported
    f = str(tmp_path / "cut.vcf")
    open(f, "w").write(src[: len(src) * 2 // 3])
    run(asan_bin, f, str(tmp_path / "out_cut"))


def test_random_mutations_never_corrupt_memory(asan_bin, tmp_path):
    rng = random.Random(20251003)
    for name in ("data3.vcf", "data5_acgt_multiallelic.vcf"):
        src = bytearray(open(os.path.join(DATA, name), "rb").read())
        extra = ("--source", "1") if "acgt" in name else ()
        for k in range(120):
            b = bytearray(src)
            for _ in range(rng.randint(1, 6)):
                op = rng.random()
                i = rng.randrange(len(b))
                if op < 0.5:
                    b[i] = rng.choice(b"\t\n:|/.,0123456789ACGT<>*#=;-e")
                elif op < 0.75:
                    del b[i:i + rng.randint(1, 20)]
                else:
                    b[i:i] = bytes(rng.choice(b"\t\n:|/.,019ACGT") for _ in range(rng.randint(1, 8)))
            f = str(tmp_path / "mut.vcf")
            open(f, "wb").write(bytes(b))
            r = run(asan_bin, f, str(tmp_path / "out_mut"), extra)
            assert r.returncode in (0, 1), (name, k, r.returncode, r.stderr[-500:])


def test_random_mutations_of_bcf_input(asan_bin, tmp_path):
    """the BCF reader under the sanitizers: byte mutations of valid raw BCF files (the host program's own truth output)"""
    rng = random.Random(7)
    r = subprocess.run([asan_bin, "-i", os.path.join(DATA, "data5_acgt_multiallelic.vcf"), "-o", str(tmp_path / "seed"), "-O", "u", "--source", "1",
                        "--seed", "1", "--depth", "inf", "-e", "0", "-printTruth", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-2000:]
    src = bytearray(open(str(tmp_path / "seed") + ".truth.bcf", "rb").read())
    hdr_end = 9 + int.from_bytes(src[5:9], "little")
    for k in range(200):
        b = bytearray(src)
        for _ in range(rng.randint(1, 4)):
            i = rng.randrange(5, len(b)) if rng.random() < 0.3 else rng.randrange(hdr_end, len(b))   # mostly inside the records
            if rng.random() < 0.7:
                b[i] = rng.randrange(256)
            else:
                del b[i:i + rng.randint(1, 9)]
        f = str(tmp_path / "mut.bcf")
        open(f, "wb").write(bytes(b))
        r = run(asan_bin, f, str(tmp_path / "out_mutb"), ("--source", "1"))
        assert r.returncode in (0, 1), (k, r.returncode, r.stderr[-500:])
