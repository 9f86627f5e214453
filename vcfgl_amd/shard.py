"""Site sharding across the GPUs of one node.

Sites are independent under VGL_RNG_TILE addressing (every value depends only on the absolute site index), so rank r
simulates a contiguous site range and nothing is exchanged during simulation.  What crosses xGMI is the end-of-run
gather of per-rank RECORDS to the writer rank -- the stand-in of the reference's per-record `bcf_write`
(vcfgl.cpp:167-206, record loop :1456-1639) for a job whose records were produced on several GPUs:

  pack_records()     a tile's kept sites (site_status >= 0) as variable-length records: per site only the
                     nGenotypes(site) / nAlleles(site) valid planes of each FORMAT tag, skipped sites dropped
                     (torch ops on the tile's own device: an HBM-bound gather, no host round trip)
  gather_records()   sizes first (one small all_gather), then point-to-point send / recv of the exact payloads to the
                     writer rank: every peer uses its own xGMI link to the writer, nothing is padded.  Backend-agnostic
                     (RCCL on GPUs: torch.distributed backend "nccl"; gloo in the CPU tests)
  unpack_records()   the writer's view: dense arrays of the kept sites, missing-filled like the tile layout

plus the two small collectives of round 1: the per-site record index (gather_site_index) and the run summary's site
counters (reduce_site_counters, vcfgl.cpp:1633).  gVCF blocks that straddle a shard boundary are stitched on the writer
(vcfgl_amd/gvcf.py)."""
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import _abi

_KIND = {f: k for f, _, k in _abi.TILE_FIELDS}
_DTYPE = {f: getattr(torch, d) for f, d, _ in _abi.TILE_FIELDS}


def site_range(rank: int, world: int, n_sites: int) -> Tuple[int, int]:
    """Contiguous split [begin, end) that keeps gVCF blocks local; remainders go to the low ranks."""
    base, rem = divmod(n_sites, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def reduce_site_counters(status: torch.Tensor, world: int, always_collective: bool = False, group=None) -> torch.Tensor:
    """[sites simulated, sites written, sites skipped] summed over the ranks: the run summary the reference prints at the
    end (vcfgl.cpp:1633) -- one small all-reduce, every rank gets the totals."""
    import torch.distributed as dist
    c = torch.stack([torch.tensor(status.numel(), device=status.device), (status >= 0).sum(), (status < 0).sum()]).to(torch.int64)
    if world > 1 or always_collective:
        dist.all_reduce(c, group=group)
    return c


def gather_site_index(status: torch.Tensor, n_alleles: torch.Tensor, world: int, rank: int,
                      n_sites_total: int, dst: int = 0, always_collective: bool = False, group=None) -> Optional[torch.Tensor]:
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
    dist.gather(idx, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        b, e = site_range(r, world, n_sites_total)
        parts.append(bufs[r][: e - b])
    return torch.cat(parts, dim=0)


# ------------------------------------------------------------------------------------------------------------------
# records

@dataclass
class PackedRecords:
    """The kept sites of one tile (or of a rank's whole range) as variable-length records.

    site0      absolute index of the tile's first site
    index      int32 [n_kept, 3]: (site index relative to site0, site_status, n_alleles)
    per_site   {field: [n_kept, ...]} per-site vectors (alleles2acgt, info_*, qs, i16 ...) of the kept sites
    per_eval   {field: [n_kept, N]}   FORMAT tags with one value per sample (DP)
    planes     {field: [rows, N]}     FORMAT tags with one plane per genotype / allele: the nG(site) (GL, PL, GP) or
                                      nA(site) (AD, ADF, ADR) valid planes of each kept site, concatenated in site order
    """
    site0: int
    n_samples: int
    index: torch.Tensor
    per_site: Dict[str, torch.Tensor] = field(default_factory=dict)
    per_eval: Dict[str, torch.Tensor] = field(default_factory=dict)
    planes: Dict[str, torch.Tensor] = field(default_factory=dict)

    @property
    def n_kept(self) -> int:
        return int(self.index.shape[0])

    def tensors(self) -> List[Tuple[str, torch.Tensor]]:
        """every tensor in a fixed order (the wire order of gather_records)"""
        out = [("index", self.index)]
        for grp in (self.per_site, self.per_eval, self.planes):
            out += [(k, grp[k]) for k in sorted(grp)]
        return out

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for _, t in self.tensors())


def _rows_per_site(field_name: str, n_alleles: torch.Tensor) -> torch.Tensor:
    na = n_alleles.to(torch.int64)
    return na * (na + 1) // 2 if _KIND[field_name] == "planeG" else na


def _pack_records_device(tile: Dict[str, torch.Tensor], site0: int) -> PackedRecords:
    """pack_records() for a tile in device memory: the library's own kernels (vgl_pack_plan_device + vgl_pack_records_device, csrc/vgl_pack.hip:
    one prefix sum over the sites, then coalesced row copies) on the current stream of the tile's device.  No fallback: without the library this raises."""
    import ctypes as C
    lib = _abi.load_library()
    status, n_alleles = tile["site_status"], tile["n_alleles"]
    dev = status.device
    S = int(status.shape[0])
    if status.dtype != torch.int32 or n_alleles.dtype != torch.int32 or not status.is_contiguous() or not n_alleles.is_contiguous():
        raise ValueError("site_status / n_alleles must be contiguous int32 tensors")
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    dev_index = dev.index if dev.index is not None else torch.cuda.current_device()

    def check(rc):
        if rc != 0:
            raise RuntimeError(f"libvcfgl_hip: {lib.vgl_last_error().decode(errors='replace')} (code {rc})")

    offsets = torch.empty((3, S + 1), dtype=torch.int32, device=dev)
    plan = _abi.PackPlan()
    check(lib.vgl_pack_plan_device(dev_index, S, status.data_ptr(), n_alleles.data_ptr(), offsets.data_ptr(), C.byref(plan), stream))
    totals = {_abi.VGL_PACK_ROW: int(plan.n_kept), _abi.VGL_PACK_ROWS_G: int(plan.rows_g), _abi.VGL_PACK_ROWS_A: int(plan.rows_a)}
    index = torch.empty((totals[_abi.VGL_PACK_ROW], 3), dtype=torch.int32, device=dev)
    p = PackedRecords(site0=int(site0), n_samples=0, index=index)
    descr = []
    N = 0
    for name, t in tile.items():
        if name in ("site_status", "n_alleles") or t is None:
            continue
        if not t.is_contiguous():
            raise ValueError(f"{name}: the tile's arrays must be contiguous")
        kind = _KIND[name]
        if kind in ("site", "site5", "siteA", "site16", "eval"):
            k, planes, tail = _abi.VGL_PACK_ROW, 1, tuple(t.shape[1:])
            out = torch.empty((totals[k],) + tail, dtype=t.dtype, device=dev)
            (p.per_eval if kind == "eval" else p.per_site)[name] = out
            if kind == "eval":
                N = int(t.shape[1])
            row_bytes = (int(t.numel()) // max(S, 1)) * t.element_size()
        else:                                                    # planeG / planeA: [S, K, N]
            k = _abi.VGL_PACK_ROWS_G if kind == "planeG" else _abi.VGL_PACK_ROWS_A
            planes, N = int(t.shape[1]), int(t.shape[2])
            out = torch.empty((totals[k], N), dtype=t.dtype, device=dev)
            p.planes[name] = out
            row_bytes = N * t.element_size()
        if out.numel() and S:
            descr.append(_abi.PackField(t.data_ptr(), out.data_ptr(), k, planes, row_bytes))
    p.n_samples = N
    if S:
        arr = (_abi.PackField * max(len(descr), 1))(*descr)
        check(lib.vgl_pack_records_device(dev_index, S, status.data_ptr(), n_alleles.data_ptr(), offsets.data_ptr(),
                                          index.data_ptr() if index.numel() else None, arr, len(descr), stream))
    p._keepalive = (offsets,)                                    # (the kernels read it after this function has returned)
    return p


def pack_records(tile: Dict[str, torch.Tensor], site0: int = 0) -> PackedRecords:
    """`tile`: {field name of vgl_tile_out: tensor} of one simulated tile; site_status and n_alleles are required.
    A tile in device memory is packed by the library's kernels (_pack_records_device: hand-written HIP behind the C ABI); the torch
    formulation below serves host tensors only (the CPU rehearsals of the N-rank path, where the oracle stands in for the kernels)."""
    status, n_alleles = tile["site_status"], tile["n_alleles"]
    if status.is_cuda:
        return _pack_records_device(tile, site0)
    S = int(status.shape[0])
    kept = status >= 0
    rows = kept.nonzero().squeeze(1)
    index = torch.stack([rows.to(torch.int32), status[rows], n_alleles[rows]], dim=1) if S else torch.zeros((0, 3), dtype=torch.int32, device=status.device)
    N = 0
    p = PackedRecords(site0=int(site0), n_samples=0, index=index)
    for name, t in tile.items():
        if name in ("site_status", "n_alleles") or t is None:
            continue
        kind = _KIND[name]
        if kind in ("site", "site5", "siteA", "site16"):
            p.per_site[name] = t.index_select(0, rows)
        elif kind == "eval":
            p.per_eval[name] = t.index_select(0, rows)
            N = int(t.shape[1])
        else:                                                    # planeG / planeA: [S, K, N]
            K, N = int(t.shape[1]), int(t.shape[2])
            want = torch.where(kept, _rows_per_site(name, n_alleles), torch.zeros((), dtype=torch.int64, device=t.device))
            mask = torch.arange(K, device=t.device)[None, :] < want[:, None]                  # [S, K]
            sel = mask.reshape(-1).nonzero().squeeze(1)
            p.planes[name] = t.reshape(S * K, N).index_select(0, sel)
    p.n_samples = N
    return p


def _missing_like(dtype, device):
    if dtype == torch.float32:
        return torch.tensor([_abi.FLOAT_MISSING_BITS], dtype=torch.int32, device=device).view(torch.float32)   # bcf_float_missing (a NaN payload)
    return torch.tensor([_abi.INT32_MISSING], dtype=dtype, device=device)


def unpack_records(p: PackedRecords, A: int, G: int) -> Dict[str, torch.Tensor]:
    """Dense arrays of the kept sites, in the tile layout ([n_kept, K, N] planes; planes beyond a site's nGenotypes /
    nAlleles hold what the device writes there: float-missing for GL / GP, int32-missing for PL, 0 for the allele depths)."""
    out = {"site_status": p.index[:, 1].clone(), "n_alleles": p.index[:, 2].clone(), "site_index": p.index[:, 0].to(torch.int64) + p.site0}
    out.update(p.per_site)
    out.update(p.per_eval)
    n = p.n_kept
    na = p.index[:, 2]
    for name, rows in p.planes.items():
        K = G if _KIND[name] == "planeG" else A
        N = int(rows.shape[1])
        per = _rows_per_site(name, na)
        if name in ("gl", "gp", "pl"):
            dense = _missing_like(rows.dtype, rows.device).expand(n * K, N).clone()
        else:
            dense = torch.zeros((n * K, N), dtype=rows.dtype, device=rows.device)
        mask = torch.arange(K, device=rows.device)[None, :] < per[:, None]
        dense[mask.reshape(-1).nonzero().squeeze(1)] = rows
        out[name] = dense.reshape(n, K, N)
    return out


def gather_records(p: PackedRecords, world: int, rank: int, dst: int = 0, transport: Optional[torch.device] = None,
                   always_collective: bool = False, group=None) -> Optional[List[PackedRecords]]:
    """Send this rank's packed records to `dst`; on `dst` returns the ranks' records in rank (= site) order.

    Step 1: one all_gather of the tensor shapes (a few int64 per rank).  Step 2: for every peer and tensor one
    point-to-point transfer of exactly that many bytes (`batch_isend_irecv`: RCCL groups them, each peer's bytes travel
    over its own xGMI link to the writer).  `transport`: device the bytes travel on when it differs from the tensors'
    (gloo moves host memory: transport=torch.device("cpu")).  `group`: the process group the bytes travel on (all ranks; default group
    when None) -- bench.py keeps a gloo control group beside the RCCL data group."""
    import torch.distributed as dist
    if world == 1 and not always_collective:
        return [p]
    names = [k for k, _ in p.tensors()]
    dev = transport if transport is not None else p.index.device
    shape = torch.tensor([p.site0, p.n_samples] + [d for _, t in p.tensors() for d in (t.shape[0], t.shape[1] if t.dim() > 1 else 1)],
                         dtype=torch.int64, device=dev)
    shapes = [torch.empty_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    ops, received = [], {}
    if rank == dst:
        for r in range(world):
            if r == rank:
                continue
            sh = shapes[r].tolist()
            bufs = []
            for k, (name, t) in enumerate(p.tensors()):
                d0, d1 = sh[2 + 2 * k], sh[3 + 2 * k]
                b = torch.empty((d0, d1) if t.dim() > 1 else (d0,), dtype=t.dtype, device=dev)
                bufs.append(b)
                if b.numel():
                    ops.append(dist.P2POp(dist.irecv, b, r, group=group))
            received[r] = (sh, bufs)
    else:
        for _, t in p.tensors():
            if t.numel():
                ops.append(dist.P2POp(dist.isend, t.contiguous().to(dev), dst, group=group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if rank != dst:
        return None
    out = []
    for r in range(world):
        if r == rank:
            out.append(p)
            continue
        sh, bufs = received[r]
        q = PackedRecords(site0=int(sh[0]), n_samples=int(sh[1]), index=bufs[0])
        for name, b in zip(names[1:], bufs[1:]):
            grp = q.per_site if name in p.per_site else q.per_eval if name in p.per_eval else q.planes
            grp[name] = b
        out.append(q)
    return out
