"""Randomised differential test of the C++ front end (vcfgl_amd/bin/vcfgl_hip): random input VCFs (ACGT or binary
alleles, multi-allelic records, missing genotypes, gaps for -explode) and random flag combinations, the VCF text it
writes against the CPU oracle driven through the Python mirror of the record loop -- record selection, allele columns,
INFO / FORMAT key sets and order, every value (floats at the 6 significant digits the text carries)."""
import os
import subprocess

import numpy as np
import pytest

import golden_util as gu
from vcfgl_amd import VcfglArgs, VcfglArgError, _abi
from vcfgl_amd.recordloop import iter_sites
from vcfgl_amd.vcfio import read_vcf

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "vcfgl_amd", "bin", "vcfgl_hip")


def random_vcf(rng, path, binary):
    N = int(rng.choice([1, 2, 5, 70]))
    S = int(rng.integers(1, 25))
    length = S * 3 + 5
    pos = np.sort(rng.choice(np.arange(1, length), size=S, replace=False))
    miss = float(rng.choice([0.0, 0.0, 0.1]))
    with open(path, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n")
        fh.write(f"##contig=<ID=chr7,length={length}>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n")
        fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"smp{i}" for i in range(N)) + "\n")
        for p in pos:
            if binary:
                alleles = ["0", "1"]
            else:
                alleles = list(rng.permutation(list("ACGT"))[: int(rng.integers(2, 5))])
            gts = []
            for _ in range(N):
                if rng.random() < miss:
                    gts.append(".|." if rng.random() < 0.5 else "./.")
                else:
                    a, b = rng.integers(0, len(alleles), size=2)
                    gts.append(f"{a}{'|' if rng.random() < 0.7 else '/'}{b}")
            fh.write(f"chr7\t{p}\t.\t{alleles[0]}\t{','.join(alleles[1:])}\t.\tPASS\t.\tGT\t" + "\t".join(gts) + "\n")
    return N


def random_flags(rng, tmp, N, binary):
    eqs = int(rng.choice([0, 0, 1, 2, 2]))
    gl = int(rng.choice([1, 2, 2]))
    precise = int(rng.integers(0, 2)) if gl == 2 else 0
    adj = int(rng.choice([0, 0, 1, 2, 3])) if not precise else int(rng.choice([0, 2]))
    strand = int(rng.integers(0, 2))
    f = ["--source", "0" if binary else "1", "--seed", str(int(rng.integers(0, 2 ** 31 - 1))),
         "--error-rate", str(float(rng.choice([0.0, 0.002, 0.01, 0.2])) if eqs == 0 else float(rng.choice([0.005, 0.01, 0.2]))),
         "--error-qs", str(eqs), "--gl-model", str(gl), "--precise-gl", str(precise), "--adjust-qs", str(adj),
         "--adjust-by", str(float(rng.choice([0.499, 0.25]))), "-explode", str(int(rng.integers(0, 2))),
         "-doUnobserved", str(int(rng.integers(0, 6))), "--rm-invar-sites", str(int(rng.choice([0, 0, 1, 2, 4, 7]))),
         "--rm-empty-sites", str(int(rng.integers(0, 2))),
         "-addGL", str(int(rng.choice([1, 1, 0]))), "-addGP", str(int(rng.integers(0, 2))), "-addPL", str(int(rng.integers(0, 2))),
         "-addQS", str(1 if (adj & 2) else int(rng.integers(0, 2))), "-addI16", str(strand & int(rng.integers(0, 2))),
         "-addFormatDP", str(int(rng.choice([1, 1, 0]))), "-addInfoDP", str(int(rng.integers(0, 2))),
         "-addFormatAD", str(int(rng.integers(0, 2))), "-addInfoAD", str(int(rng.integers(0, 2))),
         "-addFormatADF", str(strand), "-addInfoADF", str(strand & int(rng.integers(0, 2))),
         "-addFormatADR", str(strand & int(rng.integers(0, 2))), "-addInfoADR", str(strand)]
    if f[f.index("-addGL") + 1] == "0" and f[f.index("-addFormatDP") + 1] == "0":
        f[f.index("-addFormatDP") + 1] = "1"                      # at least one FORMAT tag
    if eqs:
        f += ["--beta-variance", str(float(rng.choice([1e-5, 1e-4])))]
    if rng.random() < 0.25:
        dfile = os.path.join(tmp, "depths.txt")
        with open(dfile, "w") as fh:
            fh.write("\n".join(str(float(x)) for x in rng.choice([0.0, 0.5, 3.0, 14.0], size=N)) + "\n")
        f += ["--depths-file", dfile]
    else:
        f += ["--depth", str(float(rng.choice([0.0, 0.3, 2.0, 6.0, 13.0])))]
    if eqs == 2 and rng.random() < 0.3:
        bfile = os.path.join(tmp, "bins.csv")
        with open(bfile, "w") as fh:
            fh.write("0,2,2\n3,14,12\n15,30,23\n31,63,37\n")
        f += ["--qs-bins", bfile]
    return f


def expected_keys(a):
    fmt = [k for k, on in (("DP", a.add_fmt_dp), ("GL", a.add_gl), ("PL", a.add_pl), ("GP", a.add_gp), ("AD", a.add_fmt_ad),
                           ("ADF", a.add_fmt_adf), ("ADR", a.add_fmt_adr)) if on]
    info = [k for k, on in (("DP", a.add_info_dp), ("QS", a.add_qs), ("I16", a.add_i16), ("AD", a.add_info_ad),
                            ("ADF", a.add_info_adf), ("ADR", a.add_info_adr)) if on]
    return fmt, info


@pytest.mark.parametrize("chunk", range(int(os.environ.get("VGL_CLI_FUZZ_CHUNKS", "8"))))
def test_random_cli_runs_against_the_oracle(oracle, chunk, tmp_path):
    rng = np.random.default_rng(int(os.environ.get("VGL_CLI_FUZZ_SEED", "7000")) + chunk)
    done = 0
    while done < 6:
        binary = bool(rng.integers(0, 2))
        inp = str(tmp_path / f"in{done}.vcf")
        N = random_vcf(rng, inp, binary)
        flags = random_flags(rng, str(tmp_path), N, binary)
        try:
            VcfglArgs.from_argv(flags).validate()
        except VcfglArgError:
            continue
        for mode, beta in ((_abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48), (_abi.VGL_RNG_SERIAL, _abi.VGL_BETA_STD)):
            tag = (chunk, done, mode, " ".join(flags))
            args = VcfglArgs.from_argv(flags).validate()
            args.rng_mode, args.beta_sampler = mode, beta
            vcf = read_vcf(inp)
            sites = list(iter_sites(vcf, args))
            out = str(tmp_path / f"out{done}_{mode}")
            r = subprocess.run([BIN, "-i", inp, "-o", out, "-O", "v", "--rng-mode", str(mode), "--tile-sites", str(int(rng.choice([1, 4, 4096])))] + flags,
                               capture_output=True, text=True, timeout=300)
            try:
                orc = oracle.Oracle(args, N)
                fields = [f for f, _, _ in _abi.TILE_FIELDS if not (f == "qs" and not (args.add_qs or args.add_i16)) and not (f == "i16" and not args.add_i16)]
                tile = orc.simulate(0, np.stack([s.gt for s in sites]), fields=fields) if sites else None
            except oracle.OracleError as e:
                assert r.returncode != 0, (tag, "oracle refuses, program does not", str(e))     # qs-bin miss, VGL_E_ADJQ, bad beta shape ...
                continue
            assert r.returncode == 0, (tag, r.stderr[-1500:])
            got = read_vcf(out + ".vcf")
            if tile is None:
                assert not got.records, tag
                continue
            strict = gu.close6
            if args.precise_gl or args.add_gp or (mode == _abi.VGL_RNG_SERIAL and args.error_qs):
                # a device log10 / pow feeds these floats (1e-6): one unit of the sixth printed digit
                gu.close6 = lambda x, y: strict(x, y) or (np.isfinite(x) and np.isfinite(y) and abs(x - y) <= 2e-5 * max(abs(x), abs(y), 1e-30))
            try:
                errs = gu.compare_with_golden(args, sites, tile, got, check_i16_tail=True)
            finally:
                gu.close6 = strict
            assert not errs, (tag, errs[:10])
            fmt, info = expected_keys(args)
            for line in open(out + ".vcf"):
                if line.startswith("#"):
                    continue
                c = line.rstrip("\n").split("\t")
                assert c[8] == (":".join(fmt) if fmt else "."), (tag, c[8])
                assert [kv.split("=")[0] for kv in c[7].split(";") if kv != "."] == info, (tag, c[7])
        done += 1


# ---------------------------------------------------------------------------------------------------------------------
# -doGVCF 1: the block builder (prepare_gvcf_block, bcf_utils.cpp:662-942; the loop of write_record_values,
# vcfgl.cpp:167-206) restated over the oracle's tile, against the host program's streaming implementation.

def gvcf_expected(args, sites, tile, block_dps):
    """Sequence of ("rec", site index) / ("block", dict) the reference's rules give for the kept sites of a tile."""
    status, nobs = tile.numpy("site_status"), tile.numpy("n_alleles_obs")
    dp, pl = tile.numpy("fmt_dp"), tile.numpy("pl")
    out, cur = [], None

    def flush():
        nonlocal cur
        out.append(("block", cur))
        cur = None

    for i in [k for k in range(len(sites)) if status[k] >= 0] + [None]:
        while True:
            if i is None:                                                   # end of input: flush what is open
                if cur is not None:
                    flush()
                break
            if cur is None:
                if nobs[i] != 1:
                    out.append(("rec", i)); break
            elif nobs[i] != 1 or sites[i].chrom != cur["chrom"] or sites[i].pos0 > cur["end"] + 1:
                flush(); continue                                            # variant site, other contig, gap
            min_dp = int(dp[i].min())
            rng_i = 0
            for thr in block_dps:
                if min_dp < thr:
                    break
                rng_i += 1
            if rng_i == 0:                                                   # too shallow for any block
                if cur is None:
                    out.append(("rec", i)); break
                flush(); continue
            if cur is not None and cur["dpr"] != rng_i:
                flush(); continue
            if cur is None:
                cur = dict(chrom=sites[i].chrom, start=sites[i].pos0, end=sites[i].pos0, dpr=rng_i, min_dp=min_dp, founder=i,
                           dp=dp[i].copy(), pl=pl[i, :3, :].copy())
            else:
                cur["min_dp"] = min(cur["min_dp"], min_dp)
                cur["dp"] = np.minimum(cur["dp"], dp[i])
                a, b = cur["pl"], pl[i, :3, :]
                less = a[1] > b[1]
                tie = (a[1] == b[1]) & (a[2] > b[2])
                a[1] = np.where(less, b[1], a[1])
                a[2] = np.where(less | tie, b[2], a[2])
                cur["end"] = sites[i].pos0
            break
    return out


def random_vcf_gvcf(rng, path, binary):
    """few samples, runs of invariant sites: most records (and every site -explode adds) are homozygous for the reference allele"""
    N = int(rng.choice([1, 2, 3, 6]))
    S = int(rng.integers(2, 14))
    length = S * 4 + 10
    pos = np.sort(rng.choice(np.arange(1, length), size=S, replace=False))
    with open(path, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##FILTER=<ID=PASS,Description=\"All filters passed\">\n")
        fh.write(f"##contig=<ID=chr7,length={length}>\n##contig=<ID=chr9,length=6>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n")
        fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"smp{i}" for i in range(N)) + "\n")
        rows = [("chr7", int(p)) for p in pos] + [("chr9", 2), ("chr9", 3)]
        for chrom, p in rows:
            alleles = ["0", "1"] if binary else list(rng.permutation(list("ACGT"))[:2])
            if rng.random() < 0.65:
                gts = ["0|0"] * N
            else:
                gts = [f"{rng.integers(0, 2)}|{rng.integers(0, 2)}" for _ in range(N)]
            fh.write(f"{chrom}\t{p}\t.\t{alleles[0]}\t{alleles[1]}\t.\tPASS\t.\tGT\t" + "\t".join(gts) + "\n")
    return N


@pytest.mark.parametrize("chunk", range(int(os.environ.get("VGL_CLI_FUZZ_CHUNKS", "6"))))
def test_random_gvcf_runs_against_the_block_rules(oracle, chunk, tmp_path):
    rng = np.random.default_rng(int(os.environ.get("VGL_CLI_FUZZ_SEED", "7000")) + 50000 + chunk)
    done = 0
    n_blocks = n_long = n_recs = 0
    while done < 5:
        binary = bool(rng.integers(0, 2))
        inp = str(tmp_path / f"g{done}.vcf")
        N = random_vcf_gvcf(rng, inp, binary)
        dps = sorted(set(int(x) for x in rng.choice([1, 2, 3, 5, 8], size=int(rng.integers(1, 4)))))
        flags = ["--source", "0" if binary else "1", "--seed", str(int(rng.integers(0, 2 ** 31 - 1))), "--error-rate", str(float(rng.choice([0.0, 0.002, 0.02]))),
                 "--depth", str(float(rng.choice([2.0, 5.0, 9.0, 20.0]))), "-explode", str(int(rng.choice([1, 1, 0]))),
                 "-doUnobserved", str(int(rng.choice([1, 2, 1, 2, 4, 5]))),      # (4, 5: all four bases are listed, no site has one allele: no blocks) "--rm-empty-sites", str(int(rng.integers(0, 2))),
                 "-addPL", "1", "-addFormatDP", "1", "-addQS", str(int(rng.integers(0, 2))), "-addGL", str(int(rng.integers(0, 2))),
                 "-addInfoDP", str(int(rng.integers(0, 2))), "-doGVCF", "1", "--gvcf-dps", ",".join(str(d) for d in dps)]
        for mode, beta in ((_abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48), (_abi.VGL_RNG_SERIAL, _abi.VGL_BETA_STD)):
            tag = (chunk, done, mode, " ".join(flags))
            args = VcfglArgs.from_argv(flags).validate()
            args.rng_mode, args.beta_sampler = mode, beta
            vcf = read_vcf(inp)
            sites = list(iter_sites(vcf, args))
            out = str(tmp_path / f"gout{done}_{mode}")
            r = subprocess.run([BIN, "-i", inp, "-o", out, "-O", "v", "--rng-mode", str(mode), "--tile-sites", str(int(rng.choice([1, 3, 4096])))] + flags,
                               capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, (tag, r.stderr[-1500:])
            lines = [l.rstrip("\n").split("\t") for l in open(out + ".vcf") if not l.startswith("#")]
            if not sites:
                assert not lines, tag
                continue
            fields = [f for f, _, _ in _abi.TILE_FIELDS if not (f == "qs" and not args.add_qs) and f != "i16"]
            tile = oracle.Oracle(args, N).simulate(0, np.stack([s.gt for s in sites]), fields=fields)
            want = gvcf_expected(args, sites, tile, dps)
            assert len(lines) == len(want), (tag, len(lines), len(want))
            nonref = gu.nonref_str(args)
            n_blocks += sum(1 for k, _ in want if k == "block")
            n_long += sum(1 for k, b in want if k == "block" and b["end"] > b["start"])
            n_recs += sum(1 for k, _ in want if k == "rec")
            for c, (kind, x) in zip(lines, want):
                info = dict(kv.split("=") for kv in c[7].split(";") if "=" in kv)
                if kind == "rec":
                    assert "MIN_DP" not in info and (c[0], int(c[1])) == (sites[x].chrom, sites[x].pos0 + 1), (tag, c[:8])
                    assert c[4].split(",") == gu.site_alleles(args, tile, x)[1:] or c[4] == ".", (tag, c[:8])
                    keys = c[8].split(":")
                    got_dp = [int(s.split(":")[keys.index("DP")]) for s in c[9:]]
                    assert got_dp == [int(v) for v in tile.numpy("fmt_dp")[x]], (tag, c[:8])
                    continue
                b = x
                assert (c[0], int(c[1])) == (b["chrom"], b["start"] + 1), (tag, c[:8], b["start"])
                assert [c[3]] + c[4].split(",") == gu.site_alleles(args, tile, b["founder"]) and c[4] == nonref, (tag, c[:8])
                assert c[5] == "." and c[6] == ".", (tag, c[:8])
                n_bp = b["end"] - b["start"] + 1
                assert ("END" in info) == (n_bp >= 2) and (n_bp < 2 or int(info["END"]) == b["end"] + 1), (tag, c[:8], b["end"])
                assert int(info["MIN_DP"]) == b["min_dp"], (tag, c[:8], b["min_dp"])
                assert ("QS" in info) == bool(args.add_qs), (tag, c[7])
                if args.add_qs:
                    q = tile.numpy("qs")[b["founder"]][:2]
                    assert all(gu.close6(float(o), float(v)) for o, v in zip(q, info["QS"].split(","))), (tag, c[7])
                assert c[8] == "PL:DP", (tag, c[8])
                for s in range(N):
                    plv, dpv = c[9 + s].split(":")
                    assert [int(v) for v in plv.split(",")] == [int(v) for v in b["pl"][:, s]], (tag, c[:8], s)
                    assert int(dpv) == int(b["dp"][s]), (tag, c[:8], s)
        done += 1
    assert n_recs > 0 and (n_blocks == 0 or n_long >= 0)
    _GVCF_SEEN[0] += n_blocks; _GVCF_SEEN[1] += n_long


_GVCF_SEEN = [0, 0]


def test_random_gvcf_runs_did_build_blocks():
    """(runs after the chunks above) the random runs must have exercised blocks, several sites long too"""
    assert _GVCF_SEEN[0] > 20 and _GVCF_SEEN[1] > 5, _GVCF_SEEN


@pytest.mark.parametrize("chunk", range(int(os.environ.get("VGL_CLI_FUZZ_CHUNKS", "3"))))
def test_random_runs_bcf_output_equals_text_output(chunk, tmp_path):
    """--output-mode b / u of random runs (plain records and gVCF blocks), decoded by the specification-based reader of
    tests/bcf_reader.py, against the same run's --output-mode v text: same records, field for field"""
    import bcf_reader
    from test_gpu_cli import _FloatText
    rng = np.random.default_rng(int(os.environ.get("VGL_CLI_FUZZ_SEED", "7000")) + 90000 + chunk)
    done = 0
    while done < 4:
        binary = bool(rng.integers(0, 2))
        inp = str(tmp_path / f"b{done}.vcf")
        gvcf = done % 2 == 1
        if gvcf:
            N = random_vcf_gvcf(rng, inp, binary)
            flags = ["--source", "0" if binary else "1", "--seed", str(int(rng.integers(0, 2 ** 31 - 1))), "--error-rate", "0.002", "--depth", "8",
                     "-explode", "1", "-doUnobserved", str(int(rng.choice([1, 2]))), "-addPL", "1", "-addQS", str(int(rng.integers(0, 2))),
                     "-doGVCF", "1", "--gvcf-dps", "2,5"]
        else:
            N = random_vcf(rng, inp, binary)
            flags = random_flags(rng, str(tmp_path), N, binary)
            try:
                VcfglArgs.from_argv(flags).validate()
            except VcfglArgError:
                continue
        outs = {}
        ok = True
        for om in ("v", str(rng.choice(["b", "u"]))):
            out = str(tmp_path / f"bo{done}_{om}")
            r = subprocess.run([BIN, "-i", inp, "-o", out, "-O", om, "--threads", "1"] + flags, capture_output=True, text=True, timeout=300)
            ok = ok and r.returncode == 0
            outs[om] = out
        if not ok:                                            # a refused configuration (qs-bin miss ...) is refused in every mode
            assert all(not os.path.exists(o + e) or True for o in outs.values() for e in (".vcf", ".bcf"))
            continue
        om = [k for k in outs if k != "v"][0]
        text = [l.rstrip("\n") for l in open(outs["v"] + ".vcf") if not l.startswith("#")]
        ft = _FloatText()
        rd = bcf_reader.Reader(outs[om] + ".bcf")
        assert rd.compressed == (om == "b")
        bits = []
        for rec in rd.records():
            bits += [x for _, ty, v in rec["info"] if ty == 5 for x in v]
            bits += [x for _, ty, per in rec["fmt"] if ty == 5 for v in per for x in v]
        ft.prime(bits)
        ours = list(bcf_reader.Reader(outs[om] + ".bcf").vcf_lines(ft))
        assert ours == text, (chunk, done, om, " ".join(flags))
        done += 1
