"""world_size-2 run of the N > 1 path WITH the HIP kernels: two ranks share the one GPU of the box (what `bench.py --gpus 2
--share-gpu --backend gloo` rehearses), each simulates its own site range through the C ABI into device tensors, packs the
records of its kept sites ON THE DEVICE (shard.pack_records), and the records travel to rank 0 point to point (gloo: staged
through host memory; on a multi-GPU node the same calls run on RCCL).  What arrives equals the single-process HIP run of the
whole range, field by field -- and the oracle.  (tests/test_shard_gloo.py is the same protocol with the oracle as compute.)"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIELDS = ["site_status", "n_alleles", "n_alleles_obs", "alleles2acgt", "info_dp", "info_ad", "fmt_dp", "gl", "pl", "fmt_ad"]
FLAGS = dict(seed=42, depth=0.04, error_rate=0.05, error_qs=2, beta_variance=1e-4, rm_invar_sites=4, rm_empty_sites=1, add_pl=1, add_fmt_ad=1,
             add_info_ad=1, add_info_dp=1)
N_SITES, N, SITE0 = 301, 130, 70_000


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import synth
    from vcfgl_amd import Simulator, VcfglArgs, _abi
    from vcfgl_amd.shard import gather_records, pack_records, site_range, unpack_records
    args = VcfglArgs(**FLAGS)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    dev = torch.device("cuda", 0)                                     # both ranks on the box's one GPU
    b, e = site_range(rank, world, N_SITES)
    sim = Simulator(args, N, device=0, max_sites_per_tile=e - b)
    cut = b + (e - b) // 3 + 1                                        # two tiles per rank, the first shorter
    gathered = []
    for s0, s1 in ((b, cut), (cut, e)):
        tile = sim.new_tile(s1 - s0, fields=FIELDS, device=dev)
        sim.simulate_device(SITE0 + s0, synth.binary_sites_torch(SITE0 + s0, s1 - s0, N, dev), tile)
        sim.check()
        p = pack_records({f: tile[f] for f in FIELDS}, site0=SITE0 + s0)          # on the device
        assert p.index.device.type == "cuda"
        gathered.append(gather_records(p, world, rank, transport=torch.device("cpu")))
    if rank == 0:
        dense = [{k: v.cpu().numpy() for k, v in unpack_records(p, sim.A, sim.G).items()} for per_tile in gathered for p in per_tile]
        q.put(dense)
    sim.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_gather_the_records_of_the_hip_path(oracle):
    import torch.multiprocessing as mp
    import synth
    from vcfgl_amd import Simulator, VcfglArgs, _abi
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    dense = q.get(timeout=480)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    args = VcfglArgs(**FLAGS)
    args.rng_mode, args.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    gt = synth.binary_sites(SITE0, N_SITES, N)
    sim = Simulator(args, N, device=0, max_sites_per_tile=N_SITES)
    whole = sim.simulate(SITE0, gt, fields=FIELDS)                    # the single-process HIP run
    sim.close()
    want_o = oracle.Oracle(args, N).simulate(SITE0, gt, fields=FIELDS)
    kept = whole.numpy("site_status") >= 0
    assert kept.any() and (~kept).any()
    idx = np.concatenate([d["site_index"] for d in dense])
    order = np.argsort(idx, kind="stable")
    assert np.array_equal(idx[order], SITE0 + np.nonzero(kept)[0])
    for f in FIELDS:
        got = np.concatenate([d[f] for d in dense], axis=0)[order]
        for ref in (whole, want_o):
            want = ref.numpy(f)[kept]
            assert np.array_equal(got.view(np.int32) if got.dtype == np.float32 else got, want.view(np.int32) if want.dtype == np.float32 else want), f


@pytest.mark.timeout(600)
def test_bench_two_ranks_on_one_gpu_prints_a_self_describing_line(tmp_path):
    """`bench.py --gpus 2 --share-gpu --backend gloo`: the launcher starts two ranks, both simulate their site range on the one GPU,
    the index gather / counter all-reduce run inside the timed steps and the sampled record gather after them; rank 0 prints the line
    as soon as the timed value exists and once more, as the last line, with what the sampled record gather measured -- a short line
    that says what ran (n_gpus, the process group's size and backend, the comm block); the per-rank blocks are in the detail file --
    the rehearsal of the driver's SCALE run"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--backend", "gloo", "--sites", "8192", "--tile-sites", "4096",
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--no-pack-rate", "--detail-file", str(tmp_path / "d.json")],
                       env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-1500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and all(len(l) < 4096 for l in lines)
    early, d = json.loads(lines[0]), json.loads(lines[1])
    assert early["comm"]["records_sample"] == "pending" and early["value"] == d["value"]
    assert d["n_gpus"] == 2 and d["config"]["world"] == 2 and d["config"]["rccl_world"] is None and d["config"]["backend"].startswith("gloo") and d["config"]["gather"] == "sample"
    assert d["scaling"] == "weak" and d["value"] > 0 and d["roofline"]["frac"] > 0
    c = d["comm"]
    assert c["world"] == 2 and c["gather_ms"] > 0 and c["records_sample"] == "ok" and c["records_sample_bytes_into_writer"] > 0 and c["records_sample_GBps"] > 0
    assert d["ranks"]["evals_per_s_min"] > 0 and d["ranks"]["evals_per_s_min"] <= d["ranks"]["evals_per_s_max"]
    assert abs(d["value"] - 2 * 8192 * 1000 / (d["ms_per_step"] * 1e-3)) < 1e-5 * d["value"]      # value = all ranks' evaluations / the slowest rank's time
    # every rank's own rate and kernel buckets (a multi-GPU run shows a slow rank at once): unabridged in the detail file
    full = json.load(open(tmp_path / "d.json"))
    assert "AFTER the timed steps" in full["comm"]["records_sample_note"]
    rk = full["ranks"]
    assert len(rk["evals_per_s"]["per_rank"]) == 2 and rk["evals_per_s"]["min"] > 0 and rk["evals_per_s"]["min"] <= rk["evals_per_s"]["max"]
    assert len(rk["kernel_ms_per_launch"]) == 2 and all(set(k) == {"k_depth", "k_sample", "k_redo", "k_site", "k_gl", "k_siteagg"} for k in rk["kernel_ms_per_launch"])
    assert all(k["k_sample"] > 0 and k["k_gl"] > 0 for k in rk["kernel_ms_per_launch"])
