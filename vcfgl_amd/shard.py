"""Site sharding across the GPUs of one node.  Sites are independent under VGL_RNG_TILE
addressing (every value depends only on the absolute site index), so rank r simulates a
contiguous site range and nothing is exchanged during simulation.  The single collective is the
end-of-run gather of the per-site record index to the writer rank (RCCL on GPUs, gloo in tests)."""
from typing import Optional, Tuple

import torch


def site_range(rank: int, world: int, n_sites: int) -> Tuple[int, int]:
    """Contiguous split [begin, end) that keeps gVCF blocks local; remainders go to the low ranks."""
    base, rem = divmod(n_sites, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def reduce_site_counters(status: torch.Tensor, world: int, always_collective: bool = False) -> torch.Tensor:
    """[sites simulated, sites written, sites skipped] summed over the ranks: the run summary the reference prints at the
    end (vcfgl.cpp:1633) -- one small all-reduce, every rank gets the totals."""
    import torch.distributed as dist
    c = torch.stack([torch.tensor(status.numel(), device=status.device), (status >= 0).sum(), (status < 0).sum()]).to(torch.int64)
    if world > 1 or always_collective:
        dist.all_reduce(c)
    return c


def gather_site_index(status: torch.Tensor, n_alleles: torch.Tensor, world: int, rank: int,
                      n_sites_total: int, dst: int = 0, always_collective: bool = False) -> Optional[torch.Tensor]:
    """Gather [status, n_alleles] rows of every rank's sites to `dst` in site order.
    Shards may differ by one site, so rows are padded to the largest shard."""
    import torch.distributed as dist
    if world == 1 and not always_collective:
        return torch.stack([status, n_alleles], dim=1)
    longest = -(-n_sites_total // world)
    idx = torch.full((longest, 2), -128, dtype=torch.int32, device=status.device)
    idx[: status.shape[0], 0] = status
    idx[: status.shape[0], 1] = n_alleles
    bufs = [torch.empty_like(idx) for _ in range(world)] if rank == dst else None
    dist.gather(idx, bufs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        b, e = site_range(r, world, n_sites_total)
        parts.append(bufs[r][: e - b])
    return torch.cat(parts, dim=0)
