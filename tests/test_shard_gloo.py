"""world_size-2 gloo tests of the N>1 path: each rank simulates its own contiguous site range (absolute site indexing),
packs the records of its kept sites, the RECORDS are gathered to rank 0 point to point, and what arrives equals the
single-process run field by field.  gVCF blocks straddling the shard boundary are stitched on the writer.
The compute stand-in on this GPU-less box is the CPU oracle; the collectives run on gloo (RCCL on the GPUs)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ["site_status", "n_alleles", "n_alleles_obs", "alleles2acgt", "info_dp", "info_ad", "fmt_dp", "gl", "pl", "fmt_ad"]
FLAGS = dict(seed=42, depth=3, error_rate=0.05, rm_invar_sites=4, rm_empty_sites=1, add_pl=1, add_fmt_ad=1, add_info_ad=1, add_info_dp=1)


def _tile_tensors(t, fields):
    return {f: torch.from_numpy(t.numpy(f)) for f in fields}


def _worker(rank, world, port, n_sites, N, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    import synth
    from vcfgl_amd import VcfglArgs, _abi
    from vcfgl_amd.shard import gather_records, gather_site_index, pack_records, reduce_site_counters, site_range, unpack_records
    args = VcfglArgs(**FLAGS)
    args.rng_mode = _abi.VGL_RNG_TILE
    b, e = site_range(rank, world, n_sites)
    orc = oracle_lib.Oracle(args, N)
    # two tiles per rank (the second ragged): records are packed tile by tile, as the device path does
    cut = b + (e - b) // 2 + 1
    packs = []
    for s0, s1 in ((b, cut), (cut, e)):
        t = orc.simulate(s0, synth.binary_sites(s0, s1 - s0, N), fields=FIELDS)
        packs.append((pack_records(_tile_tensors(t, FIELDS), site0=s0), t))
    gathered = [gather_records(p, world, rank) for p, _ in packs]
    t_all = orc.simulate(b, synth.binary_sites(b, e - b, N), fields=FIELDS)
    idx = gather_site_index(torch.from_numpy(t_all.numpy("site_status")), torch.from_numpy(t_all.numpy("n_alleles")), world, rank, n_sites)
    cnt = reduce_site_counters(torch.from_numpy(t_all.numpy("site_status")), world)
    if rank == 0:
        dense = []
        for per_tile in gathered:                      # tile-major here; the writer orders by site index
            for p in per_tile:
                dense.append({k: v.numpy() for k, v in unpack_records(p, orc.A, orc.G).items()})
        q.put((dense, idx.numpy(), cnt.tolist(), sum(p.nbytes() for g in gathered for p in g)))
    dist.barrier()
    dist.destroy_process_group()


def test_site_range_partition():
    from vcfgl_amd.shard import site_range
    for n in (0, 1, 7, 8, 1000, 1001):
        for w in (1, 2, 3, 8):
            r = [site_range(k, w, n) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


def test_pack_unpack_round_trip(oracle):
    """packing keeps exactly the valid planes of the kept sites and unpacking restores the tile layout"""
    import synth
    from vcfgl_amd import VcfglArgs, _abi
    from vcfgl_amd.shard import pack_records, unpack_records
    args = VcfglArgs(**FLAGS)
    args.rng_mode = _abi.VGL_RNG_TILE
    N, S = 7, 60
    t = oracle.Oracle(args, N).simulate(5, synth.binary_sites(5, S, N), fields=FIELDS)
    tile = _tile_tensors(t, FIELDS)
    p = pack_records(tile, site0=5)
    st, na = t.numpy("site_status"), t.numpy("n_alleles")
    kept = st >= 0
    assert 0 < kept.sum() < S
    assert p.n_kept == int(kept.sum())
    assert p.planes["gl"].shape[0] == int((na[kept] * (na[kept] + 1) // 2).sum()) < int(kept.sum()) * 15
    assert p.planes["fmt_ad"].shape[0] == int(na[kept].sum())
    dense = unpack_records(p, 5, 15)
    assert np.array_equal(dense["site_index"].numpy(), 5 + np.nonzero(kept)[0])
    for f in FIELDS:
        a, b = t.numpy(f)[kept], dense[f].numpy()
        assert np.array_equal(a.view(np.int32) if a.dtype == np.float32 else a, b.view(np.int32) if b.dtype == np.float32 else b), f
    empty = pack_records({k: v[:0] for k, v in tile.items()}, site0=0)             # an empty tile packs to nothing
    assert empty.n_kept == 0 and empty.nbytes() == 0


@pytest.mark.timeout(300)
def test_two_rank_gathered_records_equal_single_process(oracle):
    import oracle_lib
    import synth
    from vcfgl_amd import VcfglArgs, _abi
    n_sites, N, world = 41, 9, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_sites, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    dense, idx, cnt, nbytes = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    args = VcfglArgs(**FLAGS)
    args.rng_mode = _abi.VGL_RNG_TILE
    whole = oracle_lib.Oracle(args, N).simulate(0, synth.binary_sites(0, n_sites, N), fields=FIELDS)
    st = whole.numpy("site_status")
    kept = st >= 0
    # ---- the gathered records, ordered by site index, are the single-process run's kept sites, every field
    order = np.argsort(np.concatenate([d["site_index"] for d in dense]), kind="stable")
    assert np.array_equal(np.concatenate([d["site_index"] for d in dense])[order], np.nonzero(kept)[0])
    for f in FIELDS:
        got = np.concatenate([d[f] for d in dense], axis=0)[order]
        want = whole.numpy(f)[kept]
        assert np.array_equal(got.view(np.int32) if got.dtype == np.float32 else got, want.view(np.int32) if want.dtype == np.float32 else want), f
    assert 0 < nbytes < whole.numpy("gl").nbytes + whole.numpy("pl").nbytes + whole.numpy("fmt_ad").nbytes      # variable length: less than the padded tile
    # ---- index and counters as before
    assert np.array_equal(idx[:, 0], st) and np.array_equal(idx[:, 1], whole.numpy("n_alleles"))
    assert cnt == [n_sites, int(kept.sum()), int((~kept).sum())]
    assert (~kept).any() and kept.any()


def test_gvcf_blocks_straddling_shard_boundaries_are_stitched(oracle):
    """C5-like run (exploded hom-ref sites, -doUnobserved 2, PL): blocks built per shard and stitched equal the blocks of the
    whole range, for every split point -- including splits inside a block several sites long"""
    from vcfgl_amd import VcfglArgs, _abi, gvcf
    from vcfgl_amd.shard import site_range
    args = VcfglArgs(seed=42, depth=5.0, error_rate=0.01, do_unobserved=2, add_pl=1, do_gvcf=1)
    args.rng_mode = _abi.VGL_RNG_TILE
    N, S, dps = 3, 400, [1, 3, 6]
    gt = np.zeros((S, N), dtype=np.uint8)
    gt[rng_sites := np.random.default_rng(3).choice(S, size=25, replace=False)] = 0x10         # a few het sites break the blocks
    t = oracle.Oracle(args, N).simulate(0, gt, fields=["site_status", "n_alleles", "n_alleles_obs", "fmt_dp", "pl"])
    st, nobs, na, dp, pl = (t.numpy(f) for f in ("site_status", "n_alleles_obs", "n_alleles", "fmt_dp", "pl"))
    chrom = lambda i: "chr1" if i < 300 else "chr2"                                # a contig change inside the range too
    pos = lambda i: i if i < 300 else i - 300

    def build(b, e):
        return gvcf.build(dps, ((i, chrom(i), pos(i)) for i in range(b, e)), st[b:e], nobs[b:e], na[b:e], dp[b:e], pl[b:e])

    whole = build(0, S)
    blocks = [x for k, x in whole if k == "block"]
    assert len(blocks) > 10 and sum(1 for b in blocks if b.end > b.start) > 5 and any(k == "rec" for k, _ in whole)

    def same(a, b):
        return len(a) == len(b) and all(ka == kb and (xa == xb if ka == "rec" else xa.same(xb)) for (ka, xa), (kb, xb) in zip(a, b))

    inside = 0
    for cut in range(1, S):
        got = gvcf.stitch([build(0, cut), build(cut, S)])
        assert same(got, whole), cut
        inside += any(b.start < pos(cut) <= b.end and chrom(cut) == b.chrom and chrom(cut - 1) == b.chrom for b in blocks)
    assert inside > 50                                                             # many cuts fell inside a block
    for world in (3, 8):                                                           # shards of site_range(), chains of stitches
        parts = [build(*site_range(r, world, S)) for r in range(world)]
        assert same(gvcf.stitch(parts), whole), world
    one = np.zeros((64, N), dtype=np.uint8)                                        # one block spanning three shards
    t1 = oracle.Oracle(args, N).simulate(1000, one, fields=["site_status", "n_alleles", "n_alleles_obs", "fmt_dp", "pl"])
    a1 = [t1.numpy(f) for f in ("site_status", "n_alleles_obs", "n_alleles", "fmt_dp", "pl")]
    deep = [1]                                                                     # every depth >= 1 is one range
    b1 = lambda b, e: gvcf.build(deep, ((i, "c", i) for i in range(b, e)), *(x[b:e] for x in a1))
    w1 = b1(0, 64)
    if sum(1 for k, _ in w1 if k == "block") == 1 and len(w1) == 1:
        assert same(gvcf.stitch([b1(0, 20), b1(20, 40), b1(40, 64)]), w1)
