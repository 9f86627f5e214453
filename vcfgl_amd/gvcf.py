"""gVCF blocks on the writer side: host mirror of prepare_gvcf_block() (bcf_utils.cpp:662-942; the write loop of
write_record_values(), vcfgl.cpp:167-206) over simulated tiles, and the stitching of blocks that straddle a shard
boundary when the sites were simulated on several GPUs (flush rules bcf_utils.cpp:706-725, 785-797).

The reference feeds records one by one to a state machine: an invariant record (one observed allele) whose smallest
per-sample depth falls into a --gvcf-dps range founds or extends a block; a variant record, another contig, a gap, a
depth outside every range or a different range flushes the open block.  A block carries END, MIN_DP, the per-sample
minimum DP, the founder's alleles / QS / PL[0] and per sample the lexicographically smallest (PL[1], PL[2]) pair.

Every rank can run this machine over its own contiguous site range.  The only state that crosses a shard boundary is
the block still open at the end of a range: `stitch()` merges it with the block the next range starts with when the
serial machine would have extended it (same contig, no gap, same depth range) -- min / lexicographic-min aggregates are
associative, the founder's fields come from the left block."""
from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np


@dataclass
class Block:
    chrom: str
    start: int            # 0-based position of the first member
    end: int              # 0-based position of the last member
    dpr: int              # 1-based index of the --gvcf-dps range
    min_dp: int
    founder: int          # site index (caller's numbering) of the founding record: alleles, QS, PL[0] come from it
    dp: np.ndarray        # int32 [N] per-sample minimum depth
    pl: np.ndarray        # int32 [3, N]: the founder's PL[0]; per sample the smallest (PL[1], PL[2]) of the members

    def same(self, o: "Block") -> bool:
        return ((self.chrom, self.start, self.end, self.dpr, self.min_dp, self.founder) == (o.chrom, o.start, o.end, o.dpr, o.min_dp, o.founder)
                and np.array_equal(self.dp, o.dp) and np.array_equal(self.pl, o.pl))


Item = Tuple[str, Union[int, Block]]        # ("rec", site index) | ("block", Block)


def dp_range(min_dp: int, block_dps: Sequence[int]) -> int:
    r = 0
    for thr in block_dps:
        if min_dp < thr:
            break
        r += 1
    return r


class GvcfBuilder:
    """prepare_gvcf_block() as a streaming object: push() kept records in order, finish() at the end of input."""

    def __init__(self, block_dps: Sequence[int]):
        self.block_dps = list(block_dps)
        self.cur: Optional[Block] = None
        self.items: List[Item] = []

    def _flush(self):
        self.items.append(("block", self.cur))
        self.cur = None

    def push(self, site: int, chrom: str, pos0: int, n_obs: int, n_alleles: int, dp: np.ndarray, pl: np.ndarray):
        """dp: int32 [N]; pl: int32 [G, N] planes of the record (the first nGenotypes are valid)"""
        while True:
            cur = self.cur
            if cur is None:
                if n_obs != 1:
                    self.items.append(("rec", site)); return                        # GVCF_WRITE_SIMREC
            elif n_obs != 1 or chrom != cur.chrom or pos0 > cur.end + 1:
                self._flush(); continue                                             # variant site / other contig / gap
            min_dp = int(dp.min())
            r = dp_range(min_dp, self.block_dps)
            if r == 0:                                                              # too shallow for any block
                if cur is None:
                    self.items.append(("rec", site)); return
                self._flush(); continue
            if cur is not None and cur.dpr != r:
                self._flush(); continue
            if cur is None:
                if n_alleles != 2:
                    raise ValueError(f"an invariant gVCF record needs 2 alleles (REF + unobserved), found {n_alleles}")
                self.cur = Block(chrom, pos0, pos0, r, min_dp, site, dp.astype(np.int32).copy(), pl[:3].astype(np.int32).copy())
            else:
                cur.min_dp = min(cur.min_dp, min_dp)
                np.minimum(cur.dp, dp, out=cur.dp)
                _lexmin_into(cur.pl, pl)
                cur.end = pos0
            return

    def finish(self) -> List[Item]:
        if self.cur is not None:
            self._flush()
        return self.items


def _lexmin_into(a: np.ndarray, b: np.ndarray):
    """per sample: (a[1], a[2]) <- min((a[1], a[2]), (b[1], b[2])) lexicographically (bcf_utils.cpp:870-890)"""
    less = a[1] > b[1]
    tie = (a[1] == b[1]) & (a[2] > b[2])
    a[1] = np.where(less, b[1], a[1])
    a[2] = np.where(less | tie, b[2], a[2])


def build(block_dps: Sequence[int], sites: Iterable[Tuple[int, str, int]], status, n_obs, n_alleles, dp, pl) -> List[Item]:
    """One contiguous range: `sites` yields (site index, chrom, pos0) per row of the arrays (kept and skipped rows)."""
    b = GvcfBuilder(block_dps)
    for i, (site, chrom, pos0) in enumerate(sites):
        if status[i] < 0:
            continue
        b.push(site, chrom, pos0, int(n_obs[i]), int(n_alleles[i]), dp[i], pl[i])
    return b.finish()


def stitch(parts: Sequence[List[Item]]) -> List[Item]:
    """Items of consecutive site ranges (each built independently) -> the items of the whole job."""
    out: List[Item] = []
    for part in parts:
        part = list(part)
        if out and part and out[-1][0] == "block" and part[0][0] == "block":
            left, right = out[-1][1], part[0][1]
            if left.chrom == right.chrom and right.start <= left.end + 1 and left.dpr == right.dpr:
                m = Block(left.chrom, left.start, right.end, left.dpr, min(left.min_dp, right.min_dp), left.founder,
                          np.minimum(left.dp, right.dp), left.pl.copy())
                _lexmin_into(m.pl, right.pl)
                out[-1] = ("block", m)
                part = part[1:]
        out.extend(part)
    return out
