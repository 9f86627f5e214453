"""world_size-2 gloo test of the N>1 path: each rank simulates its own contiguous site range
(absolute site indexing), the record index is gathered to rank 0, and the result equals the
single-process run.  The compute stand-in on this GPU-less box is the CPU oracle."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_sites, N, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    import synth
    from vcfgl_amd import VcfglArgs, _abi
    from vcfgl_amd.shard import gather_site_index, reduce_site_counters, site_range
    args = VcfglArgs(seed=42, depth=3, error_rate=0.05, rm_invar_sites=4, rm_empty_sites=1)
    args.rng_mode = _abi.VGL_RNG_TILE
    b, e = site_range(rank, world, n_sites)
    t = oracle_lib.Oracle(args, N).simulate(b, synth.binary_sites(b, e - b, N), fields=["fmt_dp", "gl"])
    idx = gather_site_index(torch.from_numpy(t.numpy("site_status")), torch.from_numpy(t.numpy("n_alleles")),
                            world, rank, n_sites)
    dpsum = torch.tensor([int(t.numpy("fmt_dp").sum())])
    dist.all_reduce(dpsum)
    cnt = reduce_site_counters(torch.from_numpy(t.numpy("site_status")), world)
    if rank == 0:
        q.put((idx.numpy(), int(dpsum.item()), cnt.tolist()))
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


@pytest.mark.timeout(300)
def test_two_rank_shard_equals_single_process(oracle):
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
    idx, dpsum, cnt = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    args = VcfglArgs(seed=42, depth=3, error_rate=0.05, rm_invar_sites=4, rm_empty_sites=1)
    args.rng_mode = _abi.VGL_RNG_TILE
    whole = oracle_lib.Oracle(args, N).simulate(0, synth.binary_sites(0, n_sites, N), fields=["fmt_dp", "gl"])
    assert np.array_equal(idx[:, 0], whole.numpy("site_status"))
    assert np.array_equal(idx[:, 1], whole.numpy("n_alleles"))
    assert dpsum == int(whole.numpy("fmt_dp").sum())
    st = whole.numpy("site_status")
    assert cnt == [n_sites, int((st >= 0).sum()), int((st < 0).sum())]
    assert (idx[:, 0] < 0).any() and (idx[:, 0] == 0).any()      # both skipped and kept sites occur
