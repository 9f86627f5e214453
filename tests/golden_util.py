"""Shared helpers for the golden-VCF tests: run one of the reference's test cases through a
simulator (oracle or device library) and compare the tile with the reference's output VCF."""
import gzip
import json
import math
import os

import numpy as np

from vcfgl_amd import _abi
from vcfgl_amd.params import VcfglArgs
from vcfgl_amd.recordloop import iter_sites
from vcfgl_amd.vcfio import read_vcf

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REFVCF = os.path.join(GOLD, "ref_vcf")

with open(os.path.join(GOLD, "ref_tests.json")) as fh:
    REF_TESTS = json.load(fh)["tests"]


def load_case(name, rng_mode=_abi.VGL_RNG_SERIAL, beta_sampler=_abi.VGL_BETA_STD):
    t = REF_TESTS[name]
    args = VcfglArgs.from_argv(t["args"].split(), base_dir=os.path.join(REFVCF, "data")).validate()
    args.rng_mode, args.beta_sampler = rng_mode, beta_sampler
    vcf = read_vcf(os.path.join(REFVCF, "data", t["input"]))
    sites = list(iter_sites(vcf, args))
    gold = read_vcf(os.path.join(REFVCF, "reference", name, name + ".vcf"))
    return args, vcf, sites, gold


def nonref_str(args):
    return "<*>" if args.do_unobserved in (1, 4) else "<NON_REF>"


def site_alleles(args, tile, i):
    a2b = tile.numpy("alleles2acgt")[i]
    n = int(tile.numpy("n_alleles")[i])
    out = []
    for a in range(n):
        b = int(a2b[a])
        out.append(nonref_str(args) if b == 4 else ("ACGT"[b] if b >= 0 else "."))
    if int(tile.numpy("site_status")[i]) == _abi.VGL_SITE_NO_READS and args.do_unobserved == 0:
        out = ["."]
    return out


def close6(a, b):
    """htslib prints floats with 6 significant digits."""
    if math.isinf(a) or math.isinf(b):
        return a == b
    if a == b:
        return True
    scale = max(abs(a), abs(b))
    return abs(a - b) <= 0.51 * 10 ** (math.floor(math.log10(scale)) - 5) + 1e-30


def _fl(tok):
    return float(tok.replace("inf", "inf"))


def compare_with_golden(args, sites, tile, gold, check_i16_tail=True):
    """Returns a list of mismatch strings (empty = parity)."""
    errs = []
    N = tile.n_samples
    status = tile.numpy("site_status")
    kept = [i for i in range(len(sites)) if status[i] >= 0]
    if len(kept) != len(gold.records):
        return [f"record count: ours {len(kept)} vs golden {len(gold.records)}"]
    g_missing = np.uint32(_abi.FLOAT_MISSING_BITS)
    for i, grec in zip(kept, gold.records):
        where = f"{grec.chrom}:{grec.pos0 + 1}"
        if sites[i].chrom != grec.chrom or sites[i].pos0 != grec.pos0:
            errs.append(f"{where}: position mismatch (ours {sites[i].chrom}:{sites[i].pos0 + 1})")
            continue
        alleles = site_alleles(args, tile, i)
        gal = grec.alleles if grec.alleles != ["."] else ["."]
        if alleles != gal:
            errs.append(f"{where}: alleles ours {alleles} golden {gal}")
            continue
        nA = int(tile.numpy("n_alleles")[i])
        nObs = int(tile.numpy("n_alleles_obs")[i])
        nG = nA * (nA + 1) // 2
        # ---- INFO
        for key, vals in grec.info.items():
            if key == "DP":
                if int(vals[0]) != int(tile.numpy("info_dp")[i]):
                    errs.append(f"{where}: INFO/DP ours {tile.numpy('info_dp')[i]} golden {vals[0]}")
            elif key in ("AD", "ADF", "ADR"):
                ours = tile.numpy("info_" + key.lower())[i][:len(vals)]
                if [int(v) for v in vals] != [int(x) for x in ours]:
                    errs.append(f"{where}: INFO/{key} ours {list(ours)} golden {vals}")
            elif key == "QS":
                ours = tile.numpy("qs")[i][:len(vals)]
                if not all(close6(float(o), _fl(v)) for o, v in zip(ours, vals)):
                    errs.append(f"{where}: INFO/QS ours {list(ours)} golden {vals}")
            elif key == "I16":
                ours = tile.numpy("i16")[i]
                rng = range(16) if check_i16_tail else range(12)
                bad = [k for k in rng if not close6(float(ours[k]), _fl(vals[k]))]
                if bad:
                    errs.append(f"{where}: INFO/I16 fields {bad} ours {list(ours)} golden {vals}")
        # ---- FORMAT
        for s in range(N):
            gs = grec.samples[s]
            for key, vals in gs.items():
                miss = all(v == "." for v in vals)
                if key == "DP":
                    if int(vals[0]) != int(tile.numpy("fmt_dp")[i, s]):
                        errs.append(f"{where} s{s}: DP ours {tile.numpy('fmt_dp')[i, s]} golden {vals[0]}")
                elif key in ("GL", "GP"):
                    ours = tile.numpy(key.lower())[i, :, s]
                    bits = ours.view(np.uint32)
                    if miss:
                        if not all(bits[:max(nG, 1)] == g_missing):
                            errs.append(f"{where} s{s}: {key} expected missing, ours {list(ours[:nG])}")
                    else:
                        if len(vals) != nG or not all(close6(float(o), _fl(v)) for o, v in zip(ours[:nG], vals)):
                            errs.append(f"{where} s{s}: {key} ours {list(ours[:nG])} golden {vals}")
                elif key == "PL":
                    ours = tile.numpy("pl")[i, :, s]
                    if miss:
                        if not all(ours[:max(nG, 1)] == _abi.INT32_MISSING):
                            errs.append(f"{where} s{s}: PL expected missing, ours {list(ours[:nG])}")
                    elif [int(v) for v in vals] != [int(x) for x in ours[:nG]]:
                        errs.append(f"{where} s{s}: PL ours {list(ours[:nG])} golden {vals}")
                elif key in ("AD", "ADF", "ADR"):
                    ours = tile.numpy("fmt_" + key.lower())[i, :, s][:len(vals)]
                    if [int(v) for v in vals] != [int(x) for x in ours]:
                        errs.append(f"{where} s{s}: {key} ours {list(ours)} golden {vals}")
    return errs


def read_pileup(path):
    rows = []
    with gzip.open(path, "rt") as fh:
        for line in fh:
            f = line.rstrip("\n").split("\t")
            smp = [(int(f[3 + 3 * k]), f[4 + 3 * k], f[5 + 3 * k]) for k in range((len(f) - 3) // 3)]
            rows.append((f[0], int(f[1]), f[2], smp))
    return rows
