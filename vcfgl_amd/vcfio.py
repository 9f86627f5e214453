"""Minimal VCF text reader for the record loop (neither box has htslib).  Only what
check_rec_alleles() needs: contigs (+length), samples, CHROM/POS/REF/ALT and FORMAT/GT."""
import gzip
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple


@dataclass
class VcfRecord:
    chrom: str
    pos0: int                       # 0-based, like bcf1_t::pos
    alleles: List[str]              # REF + ALTs; ALT "." => only REF (n_allele == 1)
    gts: List[Tuple[int, int]]      # allele indices per sample, -1 = missing
    info: Dict[str, List[str]]
    fmt_keys: List[str]
    samples: List[Dict[str, List[str]]]
    id: str = "."
    qual: str = "."
    filt: str = "."


@dataclass
class VcfFile:
    header_lines: List[str]
    contigs: Dict[str, Optional[int]]
    samples: List[str]
    records: List[VcfRecord]


def _parse_gt(tok: str) -> Tuple[int, int]:
    sep = "|" if "|" in tok else "/"
    parts = tok.split(sep)
    if len(parts) == 1:
        parts = parts * 2
    return tuple(-1 if p in (".", "") else int(p) for p in parts[:2])


def read_vcf(path: str) -> VcfFile:
    opener = gzip.open if path.endswith(".gz") else open
    header, contigs, samples, records = [], {}, [], []
    with opener(path, "rt") as fh:
        for line in fh:
            line = line.rstrip("\n")
            if not line:
                continue
            if line.startswith("##"):
                header.append(line)
                if line.startswith("##contig=<"):
                    body = line[len("##contig=<"):-1]
                    kv = dict(x.split("=", 1) for x in body.split(",") if "=" in x)
                    contigs[kv["ID"]] = int(kv["length"]) if "length" in kv else None
                continue
            if line.startswith("#CHROM"):
                samples = line.split("\t")[9:]
                continue
            f = line.split("\t")
            alts = [] if f[4] == "." else f[4].split(",")
            info = {}
            if f[7] != ".":
                for kv in f[7].split(";"):
                    k, _, v = kv.partition("=")
                    info[k] = v.split(",") if v else []
            fmt = f[8].split(":") if len(f) > 8 else []
            smp = []
            gts = []
            for tok in f[9:]:
                vals = tok.split(":")
                d = {k: (vals[i].split(",") if i < len(vals) else ["."]) for i, k in enumerate(fmt)}
                smp.append(d)
                gts.append(_parse_gt(vals[fmt.index("GT")]) if "GT" in fmt else (-1, -1))
            records.append(VcfRecord(f[0], int(f[1]) - 1, [f[3]] + alts, gts, info, fmt, smp, f[2], f[5], f[6]))
    return VcfFile(header, contigs, samples, records)
