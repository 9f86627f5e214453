#!/usr/bin/env python3
"""bench.py -- site-sample GL evaluations/s of the HIP hot path on N MI355X (one process per GPU).

Workload (BASELINE.json configs[2], the depth-20 configuration the metric is quoted on):
1M sites x 1000 samples, --depth 20 -e 0.01 --error-qs 2 --beta-variance 1e-5 -GL 2, default tag
surface (GL + DP, -doUnobserved 1 => G = 15).  One step = one pass of the hot path over the whole
1e9-evaluation batch, tile by tile, with the packed true genotypes already resident in HBM and the
outputs written to HBM.  Sites shard across ranks (weak scaling: every rank simulates its own
1M-site range, addressed by absolute site index); the only collective is the end-of-step gather of
the per-site records' index fields (status / allele count) to the writer rank over RCCL.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402
from vcfgl_amd import Simulator, VcfglArgs, _abi  # noqa: E402
from vcfgl_amd.shard import gather_site_index, reduce_site_counters  # noqa: E402

try:
    METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]     # the reference's headline metric, verbatim
except Exception:
    METRIC = "site-sample GL evals/s at depth 20, 1/2/4/8 MI355X; % HBM roofline"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E vendor peak (MI355X_MICROARCH.md)


WORKLOADS = {
    # BASELINE.json configs[2]: the depth-20 configuration the metric is quoted on (default)
    "c3": dict(sites=1_000_000, samples=1000, flags=dict(depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=2),
               desc="--depth 20 -e 0.01 --error-qs 2 --beta-variance 1e-5 -GL 2"),
    # BASELINE.json configs[1]
    "c2": dict(sites=10_000, samples=100, flags=dict(depth=10.0, error_rate=0.01, gl_model=1),
               desc="--depth 10 -e 0.01 -GL 1"),
    # BASELINE.json configs[3], one GPU's share of the 8-GPU job (10M sites / 8)
    "c4": dict(sites=1_250_000, samples=2000, flags=dict(depth=30.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=2,
                                                         qs_bins=[(0, 2, 2), (3, 14, 12), (15, 30, 23), (31, 40, 37)]),
               desc="--depth 30 -e 0.01 --error-qs 2 --beta-variance 1e-5 --qs-bins rta3 -GL 2"),
    # BASELINE.json configs[4]: exploded hom-ref sites, PL for the gVCF blocks.  2M of one GPU's 6.25M sites
    # (50M / 8): the tile layout keeps G = 15 planes of GL and PL, 120 B per evaluation resident in HBM
    "c5": dict(sites=2_000_000, samples=500, flags=dict(depth=5.0, error_rate=0.01, gl_model=2, do_unobserved=2, add_pl=1), homref=True,
               desc="-explode 1 -doUnobserved 2 -addPL 1 --depth 5 -e 0.01 -GL 2 (simulation part of the gVCF job)"),
    # same shape as c3 with one fixed quality score (--error-qs 0, the reference's default)
    "fixedq": dict(sites=1_000_000, samples=1000, flags=dict(depth=20.0, error_rate=0.01, gl_model=2),
                   desc="--depth 20 -e 0.01 --error-qs 0 -GL 2"),
}


def workload_args(name="c3"):
    a = VcfglArgs(seed=42, **WORKLOADS[name]["flags"])
    a.rng_mode, a.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    a._workload = name
    return a


def cpu_baseline(args, n_samples, budget_s=15.0):
    """The CPU oracle (a port of the reference algorithm, same inputs, one core) on a bounded sample."""
    import subprocess
    so = os.path.join(ROOT, "oracle", "libvgl_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libvgl_oracle.so"])
    import oracle_lib
    o = oracle_lib.Oracle(args, n_samples)
    fields = ["fmt_dp", "gl"]
    o.simulate(0, synth.binary_sites(0, 4, n_samples), fields=fields)          # warm (page in, tables)
    n0 = 200
    t0 = time.perf_counter(); o.simulate(0, synth.binary_sites(0, n0, n_samples), fields=fields); dt = time.perf_counter() - t0
    n = int(max(n0, min(200000, budget_s / (dt / n0))))
    gt = synth.binary_sites(0, n, n_samples)
    t0 = time.perf_counter(); o.simulate(0, gt, fields=fields); dt = time.perf_counter() - t0
    res = {"value": n * n_samples / dt, "unit": "site-sample GL evals/s", "cores": 1, "kind": "port",
           "sample": f"first {n} sites x {n_samples} samples of the same workload, oracle/vgl_oracle.c (gcc -O2), {dt:.1f} s"}
    # All host cores, site-sharded in tile mode (NOT reference behaviour: the reference simulates on one
    # thread; SURVEY 8d asks for this figure beside the faithful one).  One process per core, each on its
    # own site range, about 8 s each.
    try:
        cores = min(os.cpu_count() or 1, 64)
        per = int(max(256, min(n, 8.0 * res["value"] / n_samples)))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", f"{n + k * per},{per}",
                                   "--workload", args._workload, "--samples", str(n_samples)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(cores)]
        dts = [float(json.loads(p.communicate(timeout=300)[0].strip().splitlines()[-1])["dt"]) for p in procs]
        res["all_cores"] = {"value": cores * per * n_samples / max(dts), "cores": cores,
                            "note": "site-sharded tile mode, one process per core; not reference behaviour"}
    except Exception as e:                                     # the single-core figure stands on its own
        res["all_cores"] = {"error": repr(e)[:200]}
    return res


def cpu_worker(spec, args, n_samples):
    """One shard of the all-cores CPU leg: the oracle over sites [site0, site0 + n) in chunks."""
    import oracle_lib
    site0, n = (int(x) for x in spec.split(","))
    o = oracle_lib.Oracle(args, n_samples)
    fields = ["fmt_dp", "gl"]
    o.simulate(0, synth.binary_sites(0, 4, n_samples), fields=fields)
    t0 = time.perf_counter()
    for s0 in range(site0, site0 + n, 256):
        m = min(256, site0 + n - s0)
        o.simulate(s0, synth.binary_sites(s0, m, n_samples), fields=fields)
    print(json.dumps({"dt": time.perf_counter() - t0}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--sites", type=int, default=None)
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--tile-sites", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)     # internal: one shard of the all-cores CPU leg
    opt = ap.parse_args()
    if opt.cpu_worker:                                         # never touches the GPU
        cpu_worker(opt.cpu_worker, workload_args(opt.workload), opt.samples or WORKLOADS[opt.workload]["samples"])
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):        # BENCH_FORCE_DIST: exercise the collective path at world 1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    args = workload_args(opt.workload)
    wl = WORKLOADS[opt.workload]
    S = opt.sites if opt.sites is not None else wl["sites"]
    N = opt.samples if opt.samples is not None else wl["samples"]
    TS = min(opt.tile_sites, S)
    site_base = rank * S                                      # this rank's absolute site range
    sim = Simulator(args, N, device=local_rank, max_sites_per_tile=TS)
    G = sim.G

    # ---- inputs resident in HBM before the timed region
    gt = torch.empty((S, N), dtype=torch.uint8, device=dev)
    if wl.get("homref"):
        gt.zero_()
    else:
        for s0 in range(0, S, 65536):
            n = min(65536, S - s0)
            gt[s0:s0 + n] = synth.binary_sites_torch(site_base + s0, n, N, dev)
    # ---- outputs: the whole job's tag arrays stay in HBM (65 B per evaluation)
    out = {
        "site_status": torch.empty((S,), dtype=torch.int32, device=dev),
        "n_alleles": torch.empty((S,), dtype=torch.int32, device=dev),
        "alleles2acgt": torch.empty((S, 5), dtype=torch.int8, device=dev),
        "fmt_dp": torch.empty((S, N), dtype=torch.int32, device=dev),
        "gl": torch.empty((S, G, N), dtype=torch.float32, device=dev),
    }
    if args.add_pl:
        out["pl"] = torch.empty((S, G, N), dtype=torch.int32, device=dev)
    structs = []
    for s0 in range(0, S, TS):
        n = min(TS, S - s0)
        t = _abi.TileOut()
        for k, v in out.items():
            setattr(t, k, v[s0:s0 + n].data_ptr())
        structs.append((s0, n, t))
    stream = torch.cuda.Stream(device=dev)
    import ctypes as C

    def step():
        for s0, n, t in structs:
            sim._check(sim.lib.vgl_simulate_tile_device(sim.ctx, site_base + s0, n, gt[s0:s0 + n].data_ptr(), C.byref(t),
                                                        C.c_void_p(stream.cuda_stream)))
        if dist is not None:                                  # record-index gather to the writer rank
            stream.synchronize()
            gather_site_index(out["site_status"], out["n_alleles"], world, rank, S * world, always_collective=True)
            reduce_site_counters(out["site_status"], world, always_collective=True)             # the run summary's totals

    def barrier():
        stream.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(opt.warmup):
        step()
    barrier()
    sim.check(stream.cuda_stream)
    sim.timing(True)
    sim.kernel_ms(reset=True)
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    sim.check(stream.cuda_stream)
    kms, klaunch = sim.kernel_ms(reset=True)
    sim.timing(False)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    evals_total = float(S) * N * world * opt.steps
    value = evals_total / dt
    if rank == 0:
        b_eval = 1 + 4 + 4 * G + (4 * G if args.add_pl else 0)  # packed GT in + DP out + GL (+ PL) out
        names = ["k_sample", "k_site", "k_gl"]
        dom = int(np.argmax(kms))
        avg_ms = kms[dom] / max(klaunch[dom], 1)
        evals_per_launch = float(S) * N * opt.steps / max(klaunch[dom], 1)     # average over the launches, partial last tile included
        achieved = b_eval * evals_per_launch / (avg_ms * 1e-3) / 1e9
        traffic = valu_busy = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                prof = json.load(open(tpath)).get(names[dom], {})
                traffic, valu_busy = prof.get("hbm_bytes_per_launch"), prof.get("valu_busy_frac")
            except Exception:
                traffic = None
        # the box's own device-copy bandwidth (read + write bytes of a 1 GiB device-to-device copy), the
        # second denominator SURVEY 8d asks for beside the vendor peak
        src = out["gl"].view(-1)[: min(out["gl"].numel(), 1 << 28)]
        dst = torch.empty_like(src)
        dst.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            dst.copy_(src)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 4 * 2 * src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del dst
        line = {
            "metric": METRIC if opt.workload in ("c3", "fixedq") else f"site-sample GL evals/s at depth {args.depth:g}", "value": value, "unit": "site-sample GL evals/s",
            "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": dt / opt.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{opt.workload}: {S} sites x {N} samples per GPU, {wl['desc']}, "
                                   f"tags GL+DP{'+PL' if args.add_pl else ''} (G={G}), rng tile mode, rand48 beta sampler", "tile_sites": TS,
                       "parallelism": f"site-sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "copy_bw_measured": copy_gbs, "frac_of_copy_bw": achieved / copy_gbs,
                         "valu_busy_frac": valu_busy,     # from the committed PMC profile: what actually bounds this kernel
                         "avg_launch_ms": avg_ms, "algorithmic_bytes_per_eval": b_eval,
                         "kernel_ms_total": dict(zip(names, kms)), "launches": dict(zip(names, klaunch))},
        }
        if not opt.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args, N)
        print(json.dumps(line))
    sim.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
