"""The batching record loop: host-side restatement of main_simulate_record_values()
(vcfgl.cpp:1456-1639) and check_rec_alleles() (vcfgl.cpp:75-163).  The reference feeds one
record at a time to simulate_record_values(); this loop decodes every record to a row of
packed true genotypes and hands whole tiles to the device through the C ABI."""
from dataclasses import dataclass
from typing import Iterator, List, Optional

import numpy as np

from . import _abi
from .params import VcfglArgs, VcfglArgError
from .vcfio import VcfFile, VcfRecord


@dataclass
class Site:
    chrom: str
    pos0: int
    gt: np.ndarray            # uint8 [n_samples], (a1 << 4) | a0 in ACGT space, 0xF missing
    exploded: bool            # synthesised hom-ref record (-explode 1)
    in_ref: str = "N"         # REF character of the input record (pileup column 3)


def allele_char_to_int(allele: str) -> int:
    """vcfgl.cpp:20-50."""
    if len(allele) > 1:
        return 4 if allele in ("<*>", "<NON_REF>") else -1
    return {"A": 0, "C": 1, "G": 2, "T": 3}.get(allele, -1)


def check_rec_alleles(rec: VcfRecord, args: VcfglArgs, n_samples: int):
    """Returns (status, gt_bytes): status -1/-2 = skipped by --rm-invar-sites (vcfgl.cpp:149-160)."""
    n_alleles = len(rec.alleles)
    if n_alleles > 5:
        raise VcfglArgError("Multiallelic sites with more than 4 alleles are not supported.")
    rec_alleles = [-1] * 5
    for i, al in enumerate(rec.alleles):
        if args.source == 1:
            rec_alleles[i] = allele_char_to_int(al)
            if rec_alleles[i] == -1:
                raise VcfglArgError(f"Allele '{al}' at position {rec.pos0 + 1} is not a valid base.")
        else:
            x = ord(al[0]) - ord("0")
            if x not in (0, 1):
                raise VcfglArgError(f"[--source 0] Found allele '{al}' at position {rec.pos0 + 1}. Only 0 and 1 are allowed when using binary GT source.")
            rec_alleles[i] = x
    if args.source == 0 and n_alleles > 2:
        raise VcfglArgError("Multiallelic sites are not supported when using binary GT source.")
    a0 = np.full(n_samples, -1, np.int16)
    a1 = np.full(n_samples, -1, np.int16)
    allelesum = 0
    for s, (g0, g1) in enumerate(rec.gts):
        if g0 >= 0:
            allelesum += g0
            a0[s] = rec_alleles[g0]
        if g1 >= 0:
            allelesum += g1
            a1[s] = rec_alleles[g1]
    if (args.rm_invar_sites & 1) and allelesum == 0:
        return -1, None
    if args.rm_invar_sites & 2:
        for a in range(1, n_alleles):
            if a * n_samples * 2 == allelesum:
                return -2, None
    lo = np.where(a0 < 0, _abi.VGL_GT_MISSING, a0).astype(np.uint8)
    hi = np.where(a1 < 0, _abi.VGL_GT_MISSING, a1).astype(np.uint8)
    return 0, (lo | (hi << 4)).astype(np.uint8)


def iter_sites(vcf: VcfFile, args: VcfglArgs) -> Iterator[Site]:
    """Sites in simulation order, i.e. every record for which the reference would call
    simulate_record_values() and get past its input filters (vcfgl.cpp:335-338): these are
    the records that consume random draws."""
    n = len(vcf.samples)
    explode_tpl: Optional[VcfRecord] = None
    last_contig = None
    n_in_contig = 0

    def emit(rec: VcfRecord, exploded: bool):
        status, gt = check_rec_alleles(rec, args, n)
        if status < 0:
            return None
        if (args.rm_invar_sites & 3) and len(rec.alleles) == 1:      # vcfgl.cpp:335-338
            return None
        return Site(rec.chrom, rec.pos0, gt, exploded, rec.alleles[0][0])

    def blank(tpl: VcfRecord, pos0: int) -> VcfRecord:
        return VcfRecord(tpl.chrom, pos0, tpl.alleles, [(0, 0)] * n, {}, tpl.fmt_keys, [])

    rec = None
    for rec in vcf.records:
        if rec.chrom != last_contig:
            n_in_contig = 0
            last_contig = rec.chrom
        while args.explode == 1 and n_in_contig != rec.pos0:
            if explode_tpl is None:
                explode_tpl = rec                                    # bcf_copy(explode_rec, in_rec), :1490
            b = blank(explode_tpl, n_in_contig)     # keeps the template's contig id: reference quirk,
            site = emit(b, True)                    # visible in test/reference/test18 (chr22 rows inside chr23)
            n_in_contig += 1
            if site is not None:
                yield site
        site = emit(rec, False)
        n_in_contig += 1
        if site is not None:
            yield site
    if args.explode == 1 and rec is not None:
        size = vcf.contigs.get(rec.chrom)
        while size is not None and n_in_contig != size:
            if explode_tpl is None:
                explode_tpl = rec
            b = blank(explode_tpl, n_in_contig)
            site = emit(b, True)
            n_in_contig += 1
            if site is not None:
                yield site


def tiles(sites: List[Site], max_sites: int):
    """Group consecutive sites into GT tiles [n_sites][n_samples]."""
    for i in range(0, len(sites), max_sites):
        chunk = sites[i:i + max_sites]
        yield i, np.stack([s.gt for s in chunk]) if chunk else np.zeros((0, 0), np.uint8)
