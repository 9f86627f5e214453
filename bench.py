#!/usr/bin/env python3
"""bench.py -- site-sample GL evaluations/s of the HIP hot path on N MI355X (one process per GPU).

Workload (BASELINE.json configs[2], the depth-20 configuration the metric is quoted on):
1M sites x 1000 samples, --depth 20 -e 0.01 --error-qs 2 --beta-variance 1e-5 -GL 2, default tag
surface (GL + DP, -doUnobserved 1 => G = 15).  One step = one pass of the hot path over the whole
1e9-evaluation batch, tile by tile, with the packed true genotypes already resident in HBM and the
outputs written to HBM.

Multi-GPU: `python bench.py --gpus N` starts N ranks itself (one process per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in their environment, before anything in this process touches a GPU); under `torch.distributed.run` it joins
the ranks it is given and refuses to run when WORLD_SIZE differs from --gpus.  Sites shard across ranks by absolute
site index: --scaling weak (default) gives every rank its own full-size range, --scaling strong splits the workload's
sites with shard.site_range().  Collectives per step, over RCCL: the gather of the per-site record index to the writer
rank and the all-reduce of the run summary's site counters; with --gather records also the tile-by-tile gather of the
packed RECORDS of the kept sites (shard.pack_records / gather_records) -- at C3's volume (65 GB of tags per GPU and
step) that gather is bound by the writer's links and is therefore a separate, labelled mode.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

try:
    METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]     # the reference's headline metric, verbatim
except Exception:
    METRIC = "site-sample GL evals/s at depth 20, 1/2/4/8 MI355X; % HBM roofline"
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E vendor peak (MI355X_MICROARCH.md)
# instruction roofline: 256 CUs x 4 SIMDs, 2.4 GHz, one wavefront instruction per 4 cycles of a SIMD -- the float64 / three-operand / 64-bit
# integer class (MI355X_MICROARCH.md "vector-instruction ISSUE cost"; tools/valu_rates.hip measured 4.3).  Two-operand 32-bit forms issue
# faster (2.6 measured), so a kernel made of them -- the float32 pool loop of round 5 -- can read ABOVE 1.0 against this peak: the line
# says so (`valu_peak_note`) instead of pretending a second roof it has not measured.
VALU_PEAK_WAVE_INST_PER_S = 1024 * 2.4e9 / 4.0
N_SIMD = 1024                  # 256 CUs x 4
SCLK_NOMINAL_MHZ = 2400.0
DTYPE = "f64 (f32-filtered decisions, f64 fallback)"     # every result is the reference's double arithmetic; float32 only DECIDES where an explicit bound says it can
ORACLE_PIN = {"c2": "n<=3 goldens + model definition", "gl1q": "n<=3 goldens + model definition"}   # GL model 1 = htslib errmod, not in the reference tree


class ClockSampler:
    """sclk / mclk of one GPU sampled from sysfs (pp_dpm_sclk / pp_dpm_mclk of the device's PCI function: the starred level) every 20 ms
    while the timed steps run -- a box-to-box difference of the headline (round 5: 9 % on the depth-5 configuration) is then visible in the line."""

    def __init__(self, torch_device):
        self.paths, self.samples, self._stop, self._thr = None, {"sclk": [], "mclk": []}, None, None
        try:
            import torch
            pr = torch.cuda.get_device_properties(torch_device)
            bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            base = os.path.join("/sys/bus/pci/devices", bdf)
            if os.path.exists(os.path.join(base, "pp_dpm_sclk")):
                self.paths = {"sclk": os.path.join(base, "pp_dpm_sclk"), "mclk": os.path.join(base, "pp_dpm_mclk")}
        except Exception:
            self.paths = None

    @staticmethod
    def _starred(path):
        for l in open(path):
            if "*" in l:
                return float(l.split(":")[1].strip().split("M")[0])
        return None

    def _run(self):
        while not self._stop.is_set():
            for k, pth in self.paths.items():
                try:
                    v = self._starred(pth)
                    if v:
                        self.samples[k].append(v)
                except Exception:
                    pass
            self._stop.wait(0.02)

    def start(self):
        import threading
        if self.paths:
            self.samples = {"sclk": [], "mclk": []}
            self._stop = threading.Event()
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()

    def stop(self):
        if self._thr is not None:
            self._stop.set()
            self._thr.join(timeout=1.0)
            self._thr = None
        out = {"source": os.path.dirname(self.paths["sclk"]) if self.paths else "unavailable"}
        for k in ("sclk", "mclk"):
            v = sorted(self.samples[k])
            out[k + "_mhz"] = v[len(v) // 2] if v else None
            out[k + "_mhz_min"] = v[0] if v else None
        out["samples"] = len(self.samples["sclk"])
        return out


def issue_roof_entry(workload):
    """profiles/issue_roof.json (tools/isa_hist.py + the replay of the kernel's hot loop on the GPU box): SIMD cycles per vector instruction of the
    dominant kernel's instruction mix"""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "issue_roof.json"))).get(workload, {})
    except Exception:
        return {}

RTA3 = [(0, 2, 2), (3, 14, 12), (15, 30, 23), (31, 40, 37)]
WORKLOADS = {
    # BASELINE.json configs[2]: the depth-20 configuration the metric is quoted on (default)
    "c3": dict(sites=1_000_000, samples=1000, flags=dict(depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=2),
               desc="--depth 20 -e 0.01 --error-qs 2 --beta-variance 1e-5 -GL 2"),
    # BASELINE.json configs[1]
    "c2": dict(sites=10_000, samples=100, flags=dict(depth=10.0, error_rate=0.01, gl_model=1),
               desc="--depth 10 -e 0.01 -GL 1"),
    # BASELINE.json configs[3], one GPU's share of the 8-GPU job (10M sites / 8)
    "c4": dict(sites=1_250_000, samples=2000, flags=dict(depth=30.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=2, qs_bins=RTA3),
               desc="--depth 30 -e 0.01 --error-qs 2 --beta-variance 1e-5 --qs-bins rta3 -GL 2"),
    # BASELINE.json configs[4]: exploded hom-ref sites, PL for the gVCF blocks.  2M of one GPU's 6.25M sites
    # (50M / 8): the tile layout keeps G = 15 planes of GL and PL, 120 B per evaluation resident in HBM
    "c5": dict(sites=2_000_000, samples=500, flags=dict(depth=5.0, error_rate=0.01, gl_model=2, do_unobserved=2, add_pl=1), homref=True,
               desc="-explode 1 -doUnobserved 2 -addPL 1 --depth 5 -e 0.01 -GL 2 (simulation part of the gVCF job)"),
    # c5 with FORMAT/PL in one byte per value (vgl_tile_out.pl_u8, ABI 4: PL <= 255): 80 B per evaluation instead of 125
    "c5u8": dict(sites=2_000_000, samples=500, flags=dict(depth=5.0, error_rate=0.01, gl_model=2, do_unobserved=2, add_pl=1), homref=True, narrow_pl=True,
                 desc="-explode 1 -doUnobserved 2 -addPL 1 --depth 5 -e 0.01 -GL 2, PL as uint8 (pl_u8)"),
    # same shape as c3 with one fixed quality score (--error-qs 0, the reference's default)
    "fixedq": dict(sites=1_000_000, samples=1000, flags=dict(depth=20.0, error_rate=0.01, gl_model=2),
                   desc="--depth 20 -e 0.01 --error-qs 0 -GL 2"),
    # c3's flags with the other GL paths of the reference (gl_methods.cpp:233-302 / :152-231), a quarter of c3's sites
    "gl1q": dict(sites=262_144, samples=1000, flags=dict(depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=1),
                 desc="--depth 20 -e 0.01 --error-qs 2 --beta-variance 1e-5 -GL 1"),
    "precise": dict(sites=262_144, samples=1000, flags=dict(depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=2, precise_gl=1),
                    desc="--depth 20 -e 0.01 --error-qs 2 --beta-variance 1e-5 -GL 2 --precise-gl 1"),
    # c3's flags with the reference's whole optional tag surface (test/runTests.sh:287-314, test1: -addGP -addPL -addI16 -addQS -addInfoDP
    # -addFormatAD/ADF/ADR -addInfoAD/ADF/ADR): strand draws, per-base quality sums, k_siteagg, GP; a quarter of c3's sites (244 B of tags per evaluation)
    "alltags": dict(sites=262_144, samples=1000,
                    flags=dict(depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=2, add_gp=1, add_pl=1, add_i16=1, add_qs=1, add_info_dp=1,
                               add_fmt_ad=1, add_info_ad=1, add_fmt_adf=1, add_info_adf=1, add_fmt_adr=1, add_info_adr=1),
                    fields=["fmt_dp", "gl", "pl", "gp", "fmt_ad", "fmt_adf", "fmt_adr", "info_dp", "info_ad", "info_adf", "info_adr", "qs", "i16"],
                    desc="--depth 20 -e 0.01 --error-qs 2 --beta-variance 1e-5 -GL 2 -addGP 1 -addPL 1 -addI16 1 -addQS 1 -addInfoDP 1 -addFormatAD/ADF/ADR 1 -addInfoAD/ADF/ADR 1"),
    # c3's flags + -addQS -addI16 only (what a bcftools-style caller turns on): the per-base quality sums and the strand draws
    "qsi16": dict(sites=262_144, samples=1000,
                  flags=dict(depth=20.0, error_rate=0.01, error_qs=2, beta_variance=1e-5, gl_model=2, add_i16=1, add_qs=1),
                  fields=["fmt_dp", "gl", "qs", "i16"],
                  desc="--depth 20 -e 0.01 --error-qs 2 --beta-variance 1e-5 -GL 2 -addQS 1 -addI16 1"),
}
KERNELS = ["k_depth", "k_sample", "k_redo", "k_site", "k_gl", "k_siteagg"]      # vgl_ctx_kernel_ms buckets (ABI 5: VGL_T_*)


def algorithmic_bytes_per_eval(fields, G):
    """B_eval = 1 (packed GT in) + 4 (DP out) + 4 G (GL) [+ 4 G (PL), or G as pl_u8] [+ 4 G (GP)] [+ 16 per AD-type FORMAT tag];
    per-site outputs (INFO tags, QS, I16) are not per-evaluation bytes"""
    b = 1
    for f in fields:
        b += {"fmt_dp": 4, "gl": 4 * G, "pl": 4 * G, "gp": 4 * G, "pl_u8": G, "fmt_ad": 16, "fmt_adf": 16, "fmt_adr": 16}.get(f, 0)
    return b


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def workload_args(name="c3"):
    from vcfgl_amd import VcfglArgs, _abi
    a = VcfglArgs(seed=42, **WORKLOADS[name]["flags"])
    if WORKLOADS[name].get("sample_major") or os.environ.get("BENCH_SAMPLE_MAJOR"):
        a.out_layout = _abi.VGL_LAYOUT_SAMPLE_MAJOR
    a.rng_mode, a.beta_sampler = _abi.VGL_RNG_TILE, _abi.VGL_BETA_RAND48
    a._workload = name
    return a


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = a port of the reference algorithm; test infrastructure used here only as the thing timed beside)

def cpu_single_core(args, n_samples, budget_s=12.0):
    """The CPU oracle (a port of the reference algorithm, same inputs, one core) on a bounded sample."""
    import synth
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
    return {"value": n * n_samples / dt, "unit": "site-sample GL evals/s", "cores": 1, "kind": "port", "n_sites": n,
            "sample": f"first {n} sites x {n_samples} samples of the same workload, oracle/vgl_oracle.c (gcc -O2), {dt:.1f} s",
            "sampler": "rand48 beta sampler (rng.h:426-446), as the GPU path in tile mode; the reference's default build draws quality "
                       "scores from std::mt19937 + std::gamma_distribution, measured at ~2.0e5 evals/s on this configuration (BASELINE.md section 2)"}


def cpu_all_cores(res, workload, n_samples, budget_s=6.0):
    """All host cores, site-sharded in tile mode (NOT reference behaviour: the reference simulates on one thread; SURVEY 8d asks
    for this figure beside the faithful one).  One process per core, each on its own site range, about `budget_s` each."""
    try:
        cores = min(os.cpu_count() or 1, 64)
        n = res["n_sites"]
        per = int(max(256, min(n, budget_s * res["value"] / n_samples)))
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", f"{n + k * per},{per}",
                                   "--workload", workload, "--samples", str(n_samples)],
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
    import synth
    site0, n = (int(x) for x in spec.split(","))
    o = oracle_lib.Oracle(args, n_samples)
    fields = ["fmt_dp", "gl"]
    o.simulate(0, synth.binary_sites(0, 4, n_samples), fields=fields)
    t0 = time.perf_counter()
    for s0 in range(site0, site0 + n, 256):
        m = min(256, site0 + n - s0)
        o.simulate(s0, synth.binary_sites(s0, m, n_samples), fields=fields)
    print(json.dumps({"dt": time.perf_counter() - t0}))


# ---------------------------------------------------------------------------------------------------------------------
# launching the ranks

def spawn_ranks(opt):
    """`python bench.py --gpus N` outside torch.distributed.run: start one process per GPU.  Nothing in this (parent)
    process has touched a GPU: `import torch` alone does not initialise HIP, and the children are fresh interpreters."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(opt.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(opt.gpus), LOCAL_WORLD_SIZE=str(opt.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll every rank: the first one that fails ends the run at once (a rank blocked in a rendezvous or a collective would
    # otherwise hold the launcher until the process group's own timeout, minutes after the real error was printed)
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            time.sleep(0.2)
            for p in list(live):
                r = p.poll()
                if r is None:
                    continue
                live.remove(p)
                if r != 0:
                    rc = abs(r) or 1
                    break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
    sys.exit(rc)


# ---------------------------------------------------------------------------------------------------------------------

def source_sha():
    """sha1 over the kernel / host sources of the library: a committed PMC profile carries the value of the build it was taken from"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "vcfgl_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")) or f == "Makefile":             # (the Makefile: per-file compiler options are part of the build)
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def profile_entry(workload):
    """committed PMC profile of this workload's kernels (profiles/pmc_traffic.json, written by tools/pmc_summary.py)"""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        prof = json.load(open(path))
        return prof.get(workload, {})
    except Exception:
        return {}


class _HostStream:
    """stand-in of a HIP stream for the CPU rehearsal of the N-rank path (BENCH_TEST_STUB, tests/bench_stub.py)"""
    cuda_stream = 0

    def synchronize(self):
        pass


def run_workload(name, opt, env, steps, warmup, sites=None, samples=None, with_cpu=False, gather="index", early=None, strict=True):
    """K timed passes of the hot path over one workload; returns the result dictionary (rank 0) or None.
    `early(res)`: called on rank 0 as soon as the timed value exists, BEFORE anything that is allowed to fail or stall afterwards
    (the sampled record gather over RCCL): the line is on stdout by then."""
    import contextlib
    import ctypes as C
    import threading
    import numpy as np
    import torch
    import synth
    from vcfgl_amd import Simulator, _abi
    from vcfgl_amd.shard import gather_records, gather_site_index, pack_records, reduce_site_counters, site_range
    rank, world, dist, dev, local_dev = env["rank"], env["world"], env["dist"], env["dev"], env["local_dev"]
    group = env.get("group")                                     # the data-plane process group (RCCL), None = the default group
    on_gpu = dev.type == "cuda"
    args = workload_args(name)
    wl = WORKLOADS[name]
    S_wl = sites if sites is not None else wl["sites"]
    N = samples if samples is not None else wl["samples"]
    if opt.scaling == "strong":
        site_base, site_end = site_range(rank, world, S_wl)          # the workload's sites split over the ranks
        S, S_total = site_end - site_base, S_wl
    else:
        S, site_base, S_total = S_wl, rank * S_wl, S_wl * world      # every rank its own full-size range
    TS = max(1, min(opt.tile_sites, S))
    sim = (env.get("make_sim") or Simulator)(args, N, device=local_dev, max_sites_per_tile=TS)
    G = sim.G
    info = sim.info()
    log(f"{name}: rank {rank}/{world} sites [{site_base}, {site_base + S}) x {N} samples, tiles of {TS}")

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    # ---- inputs resident in HBM before the timed region
    gt = torch.empty((S, N), dtype=torch.uint8, device=dev)
    if wl.get("homref"):
        gt.zero_()
    else:
        for s0 in range(0, S, 65536):
            n = min(65536, S - s0)
            gt[s0:s0 + n] = synth.binary_sites_torch(site_base + s0, n, N, dev)
    # ---- outputs: the whole job's tag arrays stay in HBM (65 B per evaluation at C3)
    narrow_pl = bool(wl.get("narrow_pl"))
    fields = list(wl.get("fields") or (["fmt_dp", "gl"] + ((["pl_u8"] if narrow_pl else ["pl"]) if args.add_pl else [])))
    A = sim.A
    shapes = {"site": (S,), "site5": (S, 5), "siteA": (S, A), "site16": (S, 16), "eval": (S, N), "planeG": (S, G, N), "planeA": (S, A, N)}
    kinds = {f: (dt, kind) for f, dt, kind in _abi.TILE_FIELDS}
    out = {f: torch.empty(shapes[kinds[f][1]], dtype=getattr(torch, kinds[f][0]), device=dev) for f in ["site_status", "n_alleles", "alleles2acgt"] + fields}
    structs = []
    for s0 in range(0, S, TS):
        n = min(TS, S - s0)
        t = _abi.TileOut()
        for k, v in out.items():
            setattr(t, k, v[s0:s0 + n].data_ptr())
        structs.append((s0, n, t))
    stream = torch.cuda.Stream(device=dev) if on_gpu else _HostStream()
    on_stream = (lambda: torch.cuda.stream(stream)) if on_gpu else contextlib.nullcontext
    transport = torch.device("cpu") if (dist is not None and env.get("data_backend", opt.backend) == "gloo" and on_gpu) else None
    ctl_dev = torch.device("cpu") if (transport is not None or not on_gpu) else dev       # where the small control tensors live
    gathered_bytes = [0]

    def to_transport(x):
        return x if transport is None else x.to(transport)

    comm = {"gather_s": 0.0, "sample_s": 0.0, "sample_bytes": 0, "sample_packed_bytes": 0}

    def step():
        for s0, n, t in structs:
            sim._check(sim.lib.vgl_simulate_tile_device(sim.ctx, site_base + s0, n, gt[s0:s0 + n].data_ptr(), C.byref(t),
                                                        C.c_void_p(stream.cuda_stream)))
            if dist is not None and gather == "records":           # the tile's records to the writer rank, variable length
                with on_stream():
                    p = pack_records({k: v[s0:s0 + n] for k, v in out.items()}, site0=site_base + s0)
                    got = gather_records(p, world, rank, transport=transport, always_collective=True, group=group)
                    if got is not None:
                        gathered_bytes[0] += sum(q.nbytes() for q in got)
        if dist is not None:                                      # record-index gather to the writer rank + the summary's counters
            stream.synchronize()
            t_c = time.perf_counter()
            gather_site_index(to_transport(out["site_status"]), to_transport(out["n_alleles"]), world, rank, S_total if opt.scaling == "strong" else S * world,
                              always_collective=True, group=group)
            reduce_site_counters(to_transport(out["site_status"]), world, always_collective=True, group=group)
            sync()
            comm["gather_s"] += time.perf_counter() - t_c

    def sampled_record_gather():
        """the RECORD gather north_star names, on a sample and OUTSIDE the timed steps: the packed records of this rank's last tile
        travel to the writer (point to point, one xGMI link per peer), so that the link rate is on record without moving the whole
        step's 65 GB per GPU through one writer.  Twice; the second pass is the one reported."""
        s0, n, _ = structs[-1]
        if os.environ.get("BENCH_TEST_STALL_RANK") == str(rank):  # tests/test_bench_world8_cpu.py: one rank never arrives
            time.sleep(3600)
        for rep in range(2):
            # every rank first says whether its packing worked: a rank that failed (out of memory in pack_records, say) must not
            # leave its peers waiting in the barriers and transfers below until the process group times out
            p, err = None, None
            try:
                with on_stream():
                    p = pack_records({k: v[s0:s0 + n] for k, v in out.items()}, site0=site_base + s0)
                stream.synchronize()
            except Exception as e:
                err = repr(e)[:300]
            ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=ctl_dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) == 0:
                comm["sample_error"] = err or "another rank failed to pack its records: sampled gather skipped on every rank"
                return
            dist.barrier(group=group)
            sync()
            t_c = time.perf_counter()
            with on_stream():
                got = gather_records(p, world, rank, transport=transport, always_collective=True, group=group)
            stream.synchronize()
            sync()
            dist.barrier(group=group)
            comm["sample_s"] = time.perf_counter() - t_c
            comm["sample_packed_bytes"] = p.nbytes()
            comm["sample_bytes"] = sum(q.nbytes() for q in got[1:]) if got is not None else 0      # bytes that crossed a link into the writer
            del p, got

    if dist is not None and gather == "records":
        # the writer receives one packed tile from every peer at a time: refuse up front when that cannot fit beside its own arrays
        need = (world - 1) * algorithmic_bytes_per_eval(fields, G) * TS * N * 2           # receive buffers + their unpacked view
        free = torch.cuda.mem_get_info(dev)[0] if (rank == 0 and on_gpu) else (1 << 62)
        okt = torch.tensor([1 if (rank != 0 or free > need) else 0], dtype=torch.int32, device=ctl_dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN, group=group)
        if int(okt.item()) == 0:
            if rank == 0:
                log(f"--gather records: the writer needs about {need / 1e9:.1f} GB for the peers' packed tiles but has {free / 1e9:.1f} GB free: "
                    f"use a smaller --tile-sites, fewer --sites, or --gather sample")
            dist.barrier()
            if not strict:                                         # (the short record-gather leg of a default N-rank run: reported, never fatal)
                sim.close()
                raise RuntimeError(f"the writer needs about {need / 1e9:.1f} GB for the peers' packed tiles, {free / 1e9:.1f} GB free")
            sys.exit(3)

    def barrier():
        stream.synchronize()
        sync()
        if dist is not None:
            dist.barrier(group=group)
        sync()

    for _ in range(warmup):
        step()
    barrier()
    sim.check(stream.cuda_stream)
    sim.timing(True)
    sim.kernel_ms(reset=True)
    gathered_bytes[0] = 0
    for k in comm:
        comm[k] = 0
    clocks = ClockSampler(dev) if (on_gpu and rank == 0) else None
    if clocks:
        clocks.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    clk = clocks.stop() if clocks else None
    sim.check(stream.cuda_stream)
    kms, klaunch = sim.kernel_ms(reset=True)
    sim.timing(False)
    per_rank = None
    if dist is not None:
        # every rank's own clock and kernel buckets to rank 0 (the first multi-GPU run must show a load imbalance at once), then the max
        mine = torch.tensor([dt] + [float(x) for x in kms] + [float(x) for x in klaunch], dtype=torch.float64, device=ctl_dev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine, group=group)
        per_rank = [t.cpu().tolist() for t in allr]
        dt = max(r[0] for r in per_rank)

    res = None
    if rank == 0:
        evals_total = float(S_total) * N * steps
        b_eval = algorithmic_bytes_per_eval(fields, G)
        dom = int(np.argmax(kms))
        avg_ms = max(kms[dom] / max(klaunch[dom], 1), 1e-9)
        evals_per_launch = float(S) * N * steps / max(klaunch[dom], 1)     # average over the launches, partial last tile included
        achieved = b_eval * evals_per_launch / (avg_ms * 1e-3) / 1e9
        prof = profile_entry(name)
        kern = prof.get("kernels", {})
        two_per_thread = KERNELS[dom] == "k_gl" and "k_gl" not in kern and "k_gl2" in kern     # the k_gl bucket is k_gl2 (+ k_gl2_scan, k_gl_redo): half the wavefronts
        kprof = kern.get("k_gl2" if two_per_thread else KERNELS[dom], {})
        fresh = bool(prof) and prof.get("src_sha") == source_sha()   # the committed counters describe THIS build
        # wavefronts per launch of the dominant kernel (launch geometry of vgl_sample.hip / vgl_gl.hip)
        sites_per_launch = float(S) * steps / max(klaunch[dom], 1)
        waves = sites_per_launch * N / 1024.0 if KERNELS[dom] == "k_depth" else sites_per_launch * ((N + 63) // 64) * (0.5 if two_per_thread else 1.0)
        scale = sites_per_launch / float(kprof.get("sites_per_launch", sites_per_launch) or sites_per_launch)   # profile launches -> this run's
        src = f"profiles/{prof.get('source')}_pmc_summary.json (committed counters of {'this build' if fresh else 'an EARLIER build: stale'}, not measured by this run)"
        valu = None
        if kprof.get("valu_insts_per_wave"):
            a = kprof["valu_insts_per_wave"] * waves / (avg_ms * 1e-3)
            valu = {"achieved": a, "peak": VALU_PEAK_WAVE_INST_PER_S, "unit": "wavefront VALU instructions/s", "frac": a / VALU_PEAK_WAVE_INST_PER_S,
                    "valu_insts_per_wave": kprof["valu_insts_per_wave"], "waves_per_launch": waves,
                    "valu_busy_frac_pmc": kprof.get("valu_busy_frac"), "source": src, "profile_matches_build": fresh,
                    "peak_note": "1024 SIMDs x 2.4 GHz / 4 cycles per wavefront instruction (float64 / three-operand class); two-operand 32-bit forms issue in ~2.6, so float32 kernels can exceed 1.0"}
        # instruction-issue roof (round 6): SIMD cycles the kernel's vector instructions need = instructions per wavefront (committed counters) x the
        # cycles per instruction of its hot loop's mix, measured by replaying that loop on the box (profiles/issue_roof.json) / (1024 SIMDs x sclk)
        issue = None
        ir = issue_roof_entry(name)
        if kprof.get("valu_insts_per_wave") and ir.get("issue_cycles_per_inst") and ir.get("kernel", KERNELS[dom]).startswith(KERNELS[dom]):
            sclk_mhz = (clk or {}).get("sclk_mhz") or SCLK_NOMINAL_MHZ
            lp = ir.get("loop") or {}
            it, vpw = lp.get("iterations_per_wave"), kprof["valu_insts_per_wave"]
            loop_valu = (it or 0) * (lp.get("valu_per_iteration_main_path", 0) + lp.get("rare_blocks_valu_per_iteration", 0))
            if it and lp.get("floor_cycles_per_inst_outside_the_loop") and 0 < loop_valu <= vpw:
                # iterations x (the loop's replayed every-iteration path + its rare blocks) + what is outside the loop at the part's cheapest rate: a lower bound
                cyc_wave = it * (lp["simd_cycles_per_iteration_main_path"] + lp.get("rare_blocks_cycles_per_iteration", 0.0)) + (vpw - loop_valu) * lp["floor_cycles_per_inst_outside_the_loop"]
            else:
                cyc_wave = vpw * ir["issue_cycles_per_inst"]
            bound_ms = waves * cyc_wave / (N_SIMD * sclk_mhz * 1e6) * 1e3
            issue = {"issue_cycles_per_inst": ir["issue_cycles_per_inst"], "issue_cycles_per_wave": cyc_wave, "issue_bound_ms": bound_ms, "issue_frac": bound_ms / avg_ms, "sclk_mhz_used": sclk_mhz,
                     "valu_insts_per_wave": kprof["valu_insts_per_wave"], "waves_per_launch": waves, "method": ir.get("method"), "source": ir.get("source"),
                     "profile_matches_build": fresh and ir.get("src_sha") == source_sha()}
        traffic = kprof.get("hbm_bytes_per_launch") * scale if kprof.get("hbm_bytes_per_launch") else None
        traffic_frac = (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None
        busy = kprof.get("valu_busy_frac")
        if issue is not None and traffic_frac is not None:
            bound = "hbm" if traffic_frac >= issue["issue_frac"] else "valu"
            basis = f"dominant kernel: HBM traffic {traffic_frac:.2f} of peak, instruction issue {issue['issue_frac']:.2f} of its roof"
        elif traffic_frac is not None and busy is not None:
            bound = "hbm" if traffic_frac >= busy else "valu"
            basis = f"PMC of the dominant kernel: HBM traffic {traffic_frac:.2f} of peak, VALU pipes {busy:.2f} busy"
        else:
            bound, basis = ("valu" if KERNELS[dom] == "k_sample" else "hbm"), "no committed counters for this workload: by the kernel's kind"
        step_bytes = b_eval * float(S) * N
        step_gbs = step_bytes / (dt / steps) / 1e9
        res = {
            "value": evals_total / dt, "unit": "site-sample GL evals/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup,
            "workload": f"{name}: {S} sites x {N} samples per GPU, {wl['desc']}, outputs {'+'.join(fields)} (G={G}), rng tile mode, rand48 beta sampler",
            "ctx": {k: info[k] for k in ("fused", "fused_split", "sample_lean", "depth_mode", "gl_sort", "gl_wpb", "read_cap", "pool_cap", "workspace_bytes")},
            "roofline": {"bound": bound, "bound_basis": basis, "kernel": KERNELS[dom], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "bound_note": "achieved / peak / frac are the ALGORITHMIC-bytes HBM figures of the dominant kernel the contract asks for; `step` is the same "
                                       "over the whole step (all kernels); `valu` is the instruction roofline (SURVEY H7, DESIGN.md section 4)",
                         "traffic": traffic, "traffic_frac_of_peak": traffic_frac,
                         "traffic_source": src if kprof else None,
                         "step": {"achieved": step_gbs, "frac": step_gbs / HBM_PEAK_GBS, "unit": "GB/s",
                                  "note": "algorithmic bytes of one step / wall time of one step on this GPU"},
                         "valu": valu, "issue": issue, "clocks": clk, "avg_launch_ms": avg_ms, "algorithmic_bytes_per_eval": b_eval,
                         "kernel_ms_total": dict(zip(KERNELS, kms)), "launches": dict(zip(KERNELS, klaunch))},
        }
        if per_rank is not None:
            nb = len(KERNELS)
            evals_rank = float(S) * N * steps
            res["ranks"] = {
                "evals_per_s": {"min": evals_rank / max(r[0] for r in per_rank), "max": evals_rank / min(r[0] for r in per_rank),
                                "per_rank": [evals_rank / r[0] for r in per_rank]},
                "ms_per_step": {"min": min(r[0] for r in per_rank) / steps * 1e3, "max": max(r[0] for r in per_rank) / steps * 1e3},
                "kernel_ms_per_launch": [{k: (r[1 + i] / r[1 + nb + i] if r[1 + nb + i] else 0.0) for i, k in enumerate(KERNELS)} for r in per_rank],
                "note": "each rank's own wall clock over the timed steps (value uses the slowest) and its kernel buckets: equal work per rank, so a spread here is the machine, not the sharding"}
        if dist is not None:
            res["comm"] = {"backend": env.get("data_backend", opt.backend), "world": dist.get_world_size(), "gather": gather,
                           "gather_ms": comm["gather_s"] / steps * 1e3,
                           "gather_note": "per step on the writer: gather of the per-site record index (status, allele count) + all-reduce of the site counters"}
            if gather == "sample":
                res["comm"]["records_sample"] = "pending"          # replaced below; stays if the sampled gather stalls and the watchdog ends the run
        if dist is not None and gather == "records":
            res["records_gather"] = {"bytes_per_step_at_writer": gathered_bytes[0] / steps, "GBps_into_writer": gathered_bytes[0] / dt / 1e9}
        if early is not None:
            early(res)

    if dist is not None and gather == "sample":
        # AFTER the timed steps and after rank 0 has printed the line: a failure is reported in `comm`, a stall ends the run with the
        # line already on stdout (the first 8-GPU run is the first time these point-to-point transfers execute over xGMI)
        def stalled():
            # the line is already on stdout (`early`); rank 0 prints it once more, last, with records_sample "stalled" -- a hung point-to-point transfer
            # must not read as a clean run: the LINE says so.  The exit code stays 0: `value` is complete and valid at this point (no data-path
            # collective is part of it), and a driver that drops the output of a non-zero run would lose the one 8-GPU measurement to a transfer that is
            # reported beside it, not in it.  (A stall BEFORE the value exists -- the RCCL proof -- exits 14.)
            log(f"rank {rank}: sampled record gather stalled for more than {opt.comm_timeout:g} s: ending the run; the line is printed again with records_sample \"stalled\"")
            if res is not None and early is not None:
                try:
                    res["comm"]["records_sample"] = "stalled"
                    early(res)
                except Exception:
                    pass
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)
        dog = threading.Timer(opt.comm_timeout, stalled)
        dog.daemon = True
        dog.start()
        try:
            sampled_record_gather()
        except Exception as e:
            comm["sample_error"] = repr(e)[:300]
        dog.cancel()
        if res is not None:
            c = res["comm"]
            if comm.get("sample_error"):
                c["records_sample"] = "error"
                c["records_sample_error"] = comm["sample_error"]
            elif comm["sample_s"] > 0:
                gbps = comm["sample_bytes"] / comm["sample_s"] / 1e9
                full_bytes = b_eval * float(S) * N * (world - 1)            # what a full record gather would move into the writer per step (upper bound: unpacked size)
                c.update({"records_sample": "ok", "records_sample_ms": comm["sample_s"] * 1e3, "records_sample_bytes_into_writer": comm["sample_bytes"],
                          "records_sample_GBps": gbps,
                          "full_record_gather_s_per_step_at_that_rate": (full_bytes / 1e9 / gbps) if gbps > 0 else None,
                          "records_sample_note": f"packed records of one {structs[-1][1]}-site tile per rank ({comm['sample_packed_bytes'] / 1e9:.2f} GB), point to point "
                                                 "to rank 0, measured once AFTER the timed steps (not part of `value`; --gather records puts every tile's "
                                                 "records inside the timed steps)"})
    if res is not None:
        if name == opt.workload and not opt.no_pack_rate and on_gpu:
            # device-side packing of one tile's records (what precedes the gather): an HBM-bound gather
            s0, n, _ = structs[0]
            tile = {k: v[s0:s0 + n] for k, v in out.items()}
            for _ in range(2):                                   # warm: the allocator then holds blocks of the packed sizes
                p = pack_records(tile, site0=site_base + s0)
                del p
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            p = pack_records(tile, site0=site_base + s0)
            e1.record()
            torch.cuda.synchronize()
            res["record_packing"] = {"tile_sites": n, "kept_sites": p.n_kept, "packed_bytes": p.nbytes(), "ms": e0.elapsed_time(e1),
                                     "GBps": 2 * p.nbytes() / (e0.elapsed_time(e1) * 1e-3) / 1e9, "note": "read + write bytes of shard.pack_records on one tile"}
            del p
        if with_cpu and on_gpu:
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
            res["roofline"]["copy_bw_measured"] = copy_gbs
            res["roofline"]["frac_of_copy_bw"] = achieved / copy_gbs
    sim.close()
    del gt, out, structs
    if on_gpu:
        torch.cuda.empty_cache()
    return res


def host_path_rate(opt, env, workload="c3"):
    """PCIe-inclusive rate of the host-buffer entry points (vgl_simulate_tile_async / vgl_tile_wait: page-locked destination
    buffers, two tiles in flight, the copies of one tile beside the kernels of the next): never the bench value, reported
    beside it.  The synchronous call on ordinary numpy buffers is timed too.  Tags as a record loop
    would ask for them: sample-major slabs (VGL_LAYOUT_SAMPLE_MAJOR), GL + DP, and for c5 PL in one byte."""
    import ctypes as C
    import numpy as np
    import synth
    from vcfgl_amd import Simulator, _abi
    args = workload_args(workload)
    args.out_layout = _abi.VGL_LAYOUT_SAMPLE_MAJOR
    wl = WORKLOADS[workload]
    N = wl["samples"]
    TS, tiles = (16384 * 1000) // N, 8
    sim = Simulator(args, N, device=env["local_dev"], max_sites_per_tile=TS)
    lib = sim.lib
    gts = [np.zeros((TS, N), dtype=np.uint8) if wl.get("homref") else synth.binary_sites(k * TS, TS, N) for k in range(2)]
    fields = [("site_status", TS * 4), ("n_alleles", TS * 4), ("alleles2acgt", TS * 5), ("fmt_dp", TS * N * 4), ("gl", TS * sim.G * N * 4)]
    if args.add_pl:
        fields.append(("pl_u8", TS * sim.G * N))
    b = 4 + 4 * sim.G + (sim.G if args.add_pl else 0)
    # ---- page-locked destination buffers, two sets
    sets = []
    for _ in range(2):
        t = _abi.TileOut()
        keep = {}
        for name, nbytes in fields:
            ptr = lib.vgl_host_alloc(nbytes)
            assert ptr, lib.vgl_last_error()
            keep[name] = ptr
            setattr(t, name, ptr)
        sets.append((t, keep))
    tick = [C.c_int32(), C.c_int32()]

    def run(n_tiles):
        pending = None
        for k in range(n_tiles):
            sim._check(lib.vgl_simulate_tile_async(sim.ctx, k * TS, TS, gts[k & 1].ctypes.data, C.byref(sets[k & 1][0]), C.byref(tick[k & 1])))
            if pending is not None:
                sim._check(lib.vgl_tile_wait(sim.ctx, tick[pending]))
            pending = k & 1
        sim._check(lib.vgl_tile_wait(sim.ctx, tick[pending]))

    run(2)
    t0 = time.perf_counter(); run(tiles); dt = time.perf_counter() - t0
    dp_sum = int(np.ctypeslib.as_array(C.cast(sets[(tiles - 1) & 1][1]["fmt_dp"], C.POINTER(C.c_int32)), shape=(TS * N,)).sum())
    for _, keep in sets:
        for ptr in keep.values():
            lib.vgl_host_free(ptr)
    # ---- the synchronous call on pageable numpy buffers
    tile = sim.new_tile(TS, fields=[f for f, _ in fields])
    sim._check(lib.vgl_simulate_tile(sim.ctx, 0, TS, gts[0].ctypes.data, tile.byref()))
    t0 = time.perf_counter()
    for k in range(3):
        sim._check(lib.vgl_simulate_tile(sim.ctx, k * TS, TS, gts[k & 1].ctypes.data, tile.byref()))
    dts = time.perf_counter() - t0
    sim.close()
    return {"value": tiles * TS * N / dt, "unit": "site-sample GL evals/s", "GBps_over_pcie": tiles * TS * N * b / dt / 1e9,
            "mean_depth_check": dp_sum / (TS * N),
            "sync_pageable": {"value": 3 * TS * N / dts, "GBps_over_pcie": 3 * TS * N * b / dts / 1e9},
            "note": f"vgl_simulate_tile_async + vgl_tile_wait, page-locked host buffers, two tiles in flight, {tiles} tiles of {TS} sites x {N} samples, "
                    f"sample-major slabs, {'GL+PL(u8)+DP' if args.add_pl else 'GL+DP'} copied back ({b} B per evaluation); PCIe-inclusive, never the bench value"}


# ---------------------------------------------------------------------------------------------------------------------
# the line the driver parses: <= 4 KB, scalars and short strings only (a 26.5 KB line was dropped).  Everything else --
# per-workload roofline blocks of the extras, notes, per-rank kernel buckets -- goes to the detail file and to one compact
# `bench_extra` line printed BEFORE the headline.

LINE_LIMIT = 4096


def _r(x, nd=4):
    """floats to `nd` significant digits (the line is a report, not a checkpoint)"""
    if isinstance(x, float):
        return float(f"{x:.{nd}g}") if x == x and x not in (float("inf"), float("-inf")) else None
    return x


def compact_roofline(rf):
    v = rf.get("valu") or {}
    iss = rf.get("issue") or {}
    out = {"bound": rf["bound"], "kernel": rf["kernel"], "achieved": _r(rf["achieved"], 5), "peak": rf["peak"], "unit": rf["unit"], "frac": _r(rf["frac"], 4),
           "traffic": _r(rf.get("traffic"), 5), "traffic_frac_of_peak": _r(rf.get("traffic_frac_of_peak")),
           "traffic_source": (rf.get("traffic_source") or "").split(" ")[0] or None,
           "avg_launch_ms": _r(rf["avg_launch_ms"], 5), "algorithmic_bytes_per_eval": rf["algorithmic_bytes_per_eval"],
           "step_frac": _r(rf["step"]["frac"]), "step_GBps": _r(rf["step"]["achieved"], 5),
           "issue_frac": _r(iss.get("issue_frac")), "issue_cycles_per_inst": _r(iss.get("issue_cycles_per_inst")), "issue_bound_ms": _r(iss.get("issue_bound_ms"), 5),
           "valu_insts_per_wave": _r(v.get("valu_insts_per_wave"), 5),
           "sclk_mhz": (rf.get("clocks") or {}).get("sclk_mhz"), "mclk_mhz": (rf.get("clocks") or {}).get("mclk_mhz"), "sclk_mhz_min": (rf.get("clocks") or {}).get("sclk_mhz_min"),
           "profile_matches_build": v.get("profile_matches_build"),
           "kernel_ms_per_launch": {k: _r(rf["kernel_ms_total"][k] / max(rf["launches"][k], 1)) for k in rf["kernel_ms_total"] if rf["launches"][k]},
           "note": "achieved/frac = algorithmic bytes of the dominant kernel / its HIP-event time; bound valu = instruction issue: issue_frac = [pool iterations x (the loop's "
                   "instruction sequence replayed on the box, counted cycles, + its rare blocks) + instructions outside the loop x the part's cheapest rate] / (1024 SIMD x sclk) "
                   "/ launch time: a lower bound of the kernel's time (profiles/issue_roof.json, DESIGN.md section 5)"}
    if "copy_bw_measured" in rf:
        out["copy_bw_measured"] = _r(rf["copy_bw_measured"], 5)
    return out


def compact_line(main_res, extra, opt, world, metric, dist_info=None):
    """the headline object: what bench.py prints as its LAST stdout line"""
    cb = main_res.get("cpu_baseline")
    line = {
        "metric": metric, "value": main_res["value"], "unit": main_res["unit"],
        "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": _r(main_res["ms_per_step"], 6),
        "higher_is_better": True, "scaling": opt.scaling, "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "config": {"workload": main_res["workload"][:200], "tile_sites": opt.tile_sites,
                   "parallelism": f"site-sharded x{world}"},
        "roofline": compact_roofline(main_res["roofline"]),
    }
    if dist_info:
        line["config"].update(dist_info)
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"], 5), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:120],
                                "sampler": "rand48 beta (rng.h:426-446) on both sides; reference default std-beta: ~2.0e5/s (BASELINE.md)"}
        ac = cb.get("all_cores") or {}
        if "value" in ac:
            line["cpu_baseline"]["all_cores_value"] = _r(ac["value"], 5)
            line["cpu_baseline"]["all_cores"] = ac["cores"]
    if "ctx" in main_res:
        line["ctx"] = main_res["ctx"]
    if "ranks" in main_res:
        rk = main_res["ranks"]
        line["ranks"] = {"evals_per_s_min": _r(rk["evals_per_s"]["min"], 5), "evals_per_s_max": _r(rk["evals_per_s"]["max"], 5),
                         "ms_per_step_min": _r(rk["ms_per_step"]["min"], 5), "ms_per_step_max": _r(rk["ms_per_step"]["max"], 5)}
    if "comm" in main_res:
        line["comm"] = {k: (_r(v, 5) if not isinstance(v, str) else v[:160]) for k, v in main_res["comm"].items() if not k.endswith("_note")}
    if "records_gather" in main_res:
        line["records_gather"] = {k: _r(v, 5) for k, v in main_res["records_gather"].items()}
    if "record_packing" in main_res:
        line["record_packing_GBps"] = _r(main_res["record_packing"]["GBps"], 5)
    if "value_with_record_gather" in main_res:
        line["value_with_record_gather"] = _r(main_res["value_with_record_gather"], 6)
        line["record_gather_leg"] = {k: (_r(v, 5) if not isinstance(v, str) else v[:200]) for k, v in main_res["record_gather_leg"].items()}
    elif "record_gather_leg" in main_res:
        line["record_gather_leg"] = {k: (_r(v, 5) if not isinstance(v, str) else v[:200]) for k, v in main_res["record_gather_leg"].items()}
    if extra:
        line["extra"] = {k: (_r(v.get("value"), 5) if isinstance(v, dict) and "value" in v else "error") for k, v in extra.items() if v is not None}
        pins = {k: ORACLE_PIN[k] for k in extra if k in ORACLE_PIN}
        if pins:
            line["extra_oracle_pin"] = pins
        line["extra_detail"] = "line `bench_extra` above + " + os.path.relpath(detail_path(opt), ROOT)
    return line


def compact_extra(extra):
    """one short object per extra workload (printed as its own line before the headline)"""
    out = {}
    for k, v in extra.items():
        if not isinstance(v, dict):
            continue
        if "error" in v:
            out[k] = {"error": v["error"][:120]}
        elif "roofline" in v:
            rf = v["roofline"]
            out[k] = {"value": _r(v["value"], 5), "ms_per_step": _r(v["ms_per_step"], 5), "steps": v["steps"], "kernel": rf["kernel"], "bound": rf["bound"],
                      "frac": _r(rf["frac"]), "step_frac": _r(rf["step"]["frac"]), "issue_frac": _r((rf.get("issue") or {}).get("issue_frac")),
                      "avg_launch_ms": _r(rf["avg_launch_ms"], 5), "B_eval": rf["algorithmic_bytes_per_eval"], "sclk_mhz": (rf.get("clocks") or {}).get("sclk_mhz")}
            if k in ORACLE_PIN:
                out[k]["oracle_pin"] = ORACLE_PIN[k]
        else:
            out[k] = {"value": _r(v.get("value"), 5), "GBps_over_pcie": _r(v.get("GBps_over_pcie")),
                      "sync_pageable": _r((v.get("sync_pageable") or {}).get("value"), 5)}
    return {"bench_extra": out, "unit": "site-sample GL evals/s"}


def detail_path(opt):
    return opt.detail_file or os.path.join(ROOT, "gpurun_out", "bench_detail.json")


def dumps_strict(obj):
    return json.dumps(obj, allow_nan=False, separators=(",", ":"))


def emit(main_res, extra, opt, world, metric, dist_info=None, final=True):
    """rank 0: detail file, the `bench_extra` line, then the headline line (always the last thing on stdout)"""
    line = compact_line(main_res, extra, opt, world, metric, dist_info)
    text = dumps_strict(line)
    if len(text) >= LINE_LIMIT:                                # never lose the line to its size: drop the optional blocks
        for k in ("extra", "ctx", "comm", "ranks", "record_packing_GBps", "records_gather"):
            line.pop(k, None)
            text = dumps_strict(line)
            if len(text) < LINE_LIMIT:
                break
    try:
        path = detail_path(opt)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        full = dict(line, roofline=main_res["roofline"])
        for k in ("cpu_baseline", "comm", "ranks", "record_packing", "records_gather"):
            if k in main_res:
                full[k] = main_res[k]
        full["extra"] = extra
        with open(path, "w") as f:
            json.dump(full, f)
    except Exception as e:
        log(f"detail file not written: {e!r}")
    if extra and final:
        print(dumps_strict(compact_extra(extra)), flush=True)
    print(text, flush=True)
    return text


def init_groups(opt, rank, world, local_dev, on_gpu):
    """Process groups of an N-rank run.  The default group is gloo (control plane: agreement on errors, the fall-back);
    the data plane is an RCCL group (`nccl` in torch.distributed) proved by one all-reduce before anything is timed.  If RCCL
    raises on any rank, EVERY rank falls back to the gloo group for the collectives -- `value` needs no data-path collective
    (sites shard with nothing exchanged), so an N-GPU line is still produced and says which transport carried the gathers.
    A rank that stalls in the RCCL proof is ended by a watchdog with a clear message instead of hanging the launcher."""
    import threading
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if dist.get_world_size() != opt.gpus:
        sys.exit(f"bench.py: the process group has {dist.get_world_size()} ranks, --gpus says {opt.gpus}")
    info = {"world": dist.get_world_size(), "rccl_world": None, "backend": "gloo (host staging, rehearsal)", "rccl_version": None}
    if opt.backend != "nccl":
        return dist, None, "gloo", info
    err, group = None, None

    def stalled():
        log(f"rank {rank}: the RCCL proof all-reduce did not finish in {opt.comm_timeout:g} s: ending the run (no line: the timed region needs its barrier)")
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(14)
    dog = threading.Timer(opt.comm_timeout, stalled)
    dog.daemon = True
    dog.start()
    try:
        if not on_gpu:
            raise RuntimeError("no GPU on this rank")
        try:
            group = dist.new_group(backend="nccl", device_id=torch.device("cuda", local_dev))
        except TypeError:
            group = dist.new_group(backend="nccl")
        t = torch.ones(1, dtype=torch.int32, device=torch.device("cuda", local_dev))
        dist.all_reduce(t, group=group)
        torch.cuda.synchronize()
        if int(t.item()) != world:
            raise RuntimeError(f"RCCL all-reduce over {world} ranks returned {int(t.item())}")
        info["rccl_world"] = dist.get_world_size(group)
    except Exception as e:
        err = repr(e)[:200]
    ok = torch.tensor([0 if err else 1], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)                    # over gloo: every rank learns whether RCCL came up everywhere
    dog.cancel()
    try:
        v = torch.cuda.nccl.version()
        info["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, (tuple, list)) else str(v)
    except Exception:
        pass
    if int(ok.item()) == 1:
        info["backend"] = "rccl (torch.distributed nccl)"
        return dist, group, "nccl", info
    errs = [None] * world
    dist.all_gather_object(errs, err)
    first = next((f"rank {r}: {e}" for r, e in enumerate(errs) if e), "unknown")
    log(f"rank {rank}: RCCL did not come up on every rank ({first}); the collectives fall back to gloo over host memory")
    info["backend"] = "gloo (FALLBACK: RCCL failed)"
    info["rccl_error"] = first[:200]
    return dist, None, "gloo", info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--gather", choices=["index", "sample", "records"], default="sample",
                    help="N > 1: what the writer rank receives every step -- index: the per-site record index and the counters; sample (default): "
                         "also the packed records of one tile per rank; records: the packed records of every tile")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="nccl = RCCL over xGMI; gloo stages through host memory (rehearsal)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a box with fewer GPUs than ranks: rank r uses device r %% device_count (needs --backend gloo)")
    ap.add_argument("--comm-timeout", type=float, default=60.0, help="seconds a rank waits in the RCCL proof / the sampled record gather before it ends the run")
    ap.add_argument("--sites", type=int, default=None)
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--tile-sites", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other configurations attached as `extra`")
    ap.add_argument("--no-pack-rate", action="store_true")
    ap.add_argument("--no-records-leg", action="store_true", help="N > 1 with --gather sample: skip the short second leg that gathers every tile's records inside the timed step")
    ap.add_argument("--records-leg-tiles", type=int, default=4, help="tiles per rank of that leg (a budget the writer's memory and a few seconds hold)")
    ap.add_argument("--detail-file", default=None, help="where the full (unabridged) result goes; default gpurun_out/bench_detail.json")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)     # internal: one shard of the all-cores CPU leg
    ap.add_argument("--cpu-single-worker", action="store_true", help=argparse.SUPPRESS)   # internal: the one-core CPU leg in a process of its own
    ap.add_argument("--host-path-worker", default=None, help=argparse.SUPPRESS)   # internal: the PCIe-inclusive host path in a fresh process
    opt = ap.parse_args()
    if opt.cpu_worker:                                         # never touches the GPU
        cpu_worker(opt.cpu_worker, workload_args(opt.workload), opt.samples or WORKLOADS[opt.workload]["samples"])
        return
    if opt.cpu_single_worker:                                  # never touches the GPU
        print(json.dumps(cpu_single_core(workload_args(opt.workload), opt.samples or WORKLOADS[opt.workload]["samples"])))
        return
    if opt.host_path_worker is not None:
        print(json.dumps({wl: host_path_rate(opt, {"local_dev": int(opt.host_path_worker)}, wl) for wl in ("c3", "c5")}))
        return
    if opt.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "RANK" not in os.environ and opt.gpus > 1:
        spawn_ranks(opt)                                       # does not return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("BENCH_TEST_LAUNCHER"):                  # tests/test_bench_launch_cpu.py: rank 1 dies at once, the others hang
        if rank == 1:
            sys.exit("bench.py: rank 1 fails (launcher test)")
        time.sleep(600)
    if world != opt.gpus:
        sys.exit(f"bench.py: --gpus {opt.gpus} but WORLD_SIZE is {world}: launch with --nproc-per-node {opt.gpus} (or plain `python bench.py --gpus {opt.gpus}`)")
    import torch
    stub = os.environ.get("BENCH_TEST_STUB")                   # tests/test_bench_world8_cpu.py: the N-rank path on a box without GPUs
    make_sim = None
    if stub:
        import bench_stub                                      # tests/bench_stub.py: fills the tile's status arrays, simulates nothing
        make_sim, local_dev, dev = bench_stub.StubSimulator, 0, torch.device("cpu")
    else:
        ndev = torch.cuda.device_count()
        if opt.share_gpu:
            if opt.backend != "gloo":
                sys.exit("--share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
            local_dev = local_rank % max(ndev, 1)
        else:
            if local_rank >= ndev:
                sys.exit(f"bench.py: rank {rank} needs GPU {local_rank} but this node shows {ndev} device(s)")
            local_dev = local_rank
        dev = torch.device("cuda", local_dev)
        torch.cuda.set_device(local_dev)
    dist, group, data_backend, dist_info = None, None, None, None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):        # BENCH_FORCE_DIST: exercise the collective path at world 1
        dist, group, data_backend, dist_info = init_groups(opt, rank, world, local_dev, dev.type == "cuda")
        dist_info["gather"] = opt.gather
    env = {"rank": rank, "world": world, "dist": dist, "group": group, "dev": dev, "local_dev": local_dev, "make_sim": make_sim}
    if data_backend:
        env["data_backend"] = data_backend
    args = workload_args(opt.workload)
    metric = METRIC if opt.workload in ("c3", "fixedq") else f"site-sample GL evals/s at depth {args.depth:g}"
    solo = world == 1 and dist is None and not stub
    want_extra = solo and not opt.no_extra and opt.sites is None and opt.samples is None

    host_first = None
    if want_extra:
        # The PCIe-inclusive host path, in a process of its own and BEFORE the device-resident workloads: device memory that has
        # been through 100+ GB of allocation and release (by this process or an earlier one on the GPU) copies back at 35 GB/s
        # instead of 53 (measured: tools/stream_alias_probe.py) -- a record loop creates its context at start, like this.
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--host-path-worker", str(local_dev)],
                                 capture_output=True, text=True, timeout=600)
            host_first = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:
            host_first = {"error": repr(e)[:300]}

    def early(res):                                            # N > 1: the line is on stdout before the sampled record gather is attempted
        emit(res, {}, opt, world, metric, dist_info, final=False)
    main_res = run_workload(opt.workload, opt, env, opt.steps, opt.warmup, sites=opt.sites, samples=opt.samples,
                            with_cpu=(not opt.no_cpu_baseline and solo), gather=opt.gather, early=(early if dist is not None else None))
    if dist is not None and opt.gather == "sample" and not opt.no_records_leg:
        # north_star's record gather INSIDE the timed region, on a tile budget that fits the writer: `value` above moves no records (sites shard with
        # nothing exchanged; a full record gather is bound by the writer's links) -- this leg puts the figure WITH the gather beside it.
        # One warm-up and one timed pass over --records-leg-tiles tiles per rank, every tile's packed records point to point into rank 0.
        import threading
        leg_sites = min(opt.sites if opt.sites is not None else WORKLOADS[opt.workload]["sites"], max(1, opt.records_leg_tiles) * opt.tile_sites)

        def leg_stalled():
            log(f"rank {rank}: the record-gather leg stalled for more than {3 * opt.comm_timeout:g} s: ending the run; the line is printed again with record_gather_leg.status \"stalled\"")
            if rank == 0 and main_res is not None:
                try:
                    main_res["record_gather_leg"] = {"status": "stalled", "sites_per_rank": leg_sites}
                    emit(main_res, {}, opt, world, metric, dist_info, final=False)
                except Exception:
                    pass
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)                                            # (as the sampled gather's watchdog: `value` stands, the line says what stalled)
        dog = threading.Timer(3 * opt.comm_timeout, leg_stalled)
        dog.daemon = True
        dog.start()
        try:
            leg = run_workload(opt.workload, opt, env, steps=1, warmup=1, sites=leg_sites, samples=opt.samples, gather="records", strict=False)
            if rank == 0:
                rg = leg.get("records_gather", {})
                main_res["value_with_record_gather"] = leg["value"]
                main_res["record_gather_leg"] = {"status": "ok", "sites_per_rank": leg_sites, "ms_per_step": leg["ms_per_step"],
                                                 "GBps_into_writer": rg.get("GBps_into_writer"), "bytes_into_writer": rg.get("bytes_per_step_at_writer"),
                                                 "note": "every tile's packed records gathered to rank 0 inside the timed step; `value` gathers none"}
        except Exception as e:
            if rank == 0 and main_res is not None:
                main_res["record_gather_leg"] = {"status": "error", "error": repr(e)[:200], "sites_per_rank": leg_sites}
        dog.cancel()
    cpu_proc = None
    if rank == 0 and solo and not opt.no_cpu_baseline:
        # the one-core CPU leg runs in its own process BESIDE the extra GPU workloads (this thread only enqueues launches and waits);
        # the all-cores leg afterwards, with the GPU idle
        log("cpu baseline (one core) started ...")
        cpu_proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-single-worker", "--workload", opt.workload] +
                                    (["--samples", str(opt.samples)] if opt.samples else []), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    extra = {}
    if want_extra:
        # the other BASELINE configurations on this GPU, a few passes each (parity-test cases, not the headline: VERDICT r1 asked
        # for driver-timed evidence of them).  Their full blocks go to the detail file, one short object each to the `bench_extra` line.
        for name in ("c5", "c5u8", "fixedq", "c4", "c2", "gl1q", "precise", "alltags", "qsi16"):
            if name == opt.workload:
                continue
            try:
                r = run_workload(name, opt, env, steps=(20 if name == "c2" else 3), warmup=1, gather="index")
                extra[name] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "workload", "ctx", "roofline")}
            except Exception as e:                             # an extra must never cost the headline line
                extra[name] = {"error": repr(e)[:300]}
        extra["host_path_c3"] = host_first.get("c3", host_first) if isinstance(host_first, dict) else host_first
        extra["host_path_c5"] = host_first.get("c5") if isinstance(host_first, dict) else None
    if cpu_proc is not None:
        try:
            cb = json.loads(cpu_proc.communicate(timeout=600)[0].strip().splitlines()[-1])
            log("cpu baseline (all cores) ...")
            main_res["cpu_baseline"] = cpu_all_cores(cb, opt.workload, opt.samples or WORKLOADS[opt.workload]["samples"])
        except Exception as e:
            log(f"cpu baseline failed: {e!r}")

    if rank == 0:
        emit(main_res, extra, opt, world, metric, dist_info, final=True)
    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:                                 # the line is out; a teardown problem must not turn the run into a failure
            log(f"rank {rank}: process-group teardown: {e!r}")


if __name__ == "__main__":
    main()
