"""First-contact safety of `bench.py --gpus 8`, rehearsed without GPUs: eight ranks started by bench.py itself,
the device path replaced by tests/bench_stub.py (BENCH_TEST_STUB: it simulates nothing), the collectives real.  On this box RCCL
cannot come up, which is exactly one of the failures the first 8-GPU run may meet: every rank must agree to fall back to gloo, the
line must still be printed with n_gpus 8 and must say which transport carried the gathers; and a rank that never arrives in the
sampled record gather must not cost the line that is already on stdout."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
COMMON = ["--sites", "64", "--samples", "8", "--tile-sites", "32", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"]


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(BENCH_TEST_STUB="1", **kw)
    return e


def _lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def test_world8_line_schema_with_rccl_fallback(tmp_path):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--detail-file", str(tmp_path / "d.json")] + COMMON, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _lines(r.stdout)
    assert len(lines) == 2                                   # the early line (before the sampled gather) and the final one
    early, line = lines
    assert len(r.stdout.strip().splitlines()[-1]) < 4096
    assert early["comm"]["records_sample"] == "pending" and early["value"] == line["value"]
    assert line["n_gpus"] == 8 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    cfg = line["config"]
    assert cfg["rccl_world"] is None and cfg["backend"].startswith("gloo (FALLBACK") and "no GPU" in cfg["rccl_error"] and cfg["gather"] == "sample"
    assert "rccl_version" in cfg
    assert "fall back to gloo" in r.stderr
    comm = line["comm"]
    assert comm["world"] == 8 and comm["backend"] == "gloo" and comm["records_sample"] == "ok"
    # 7 peers x the packed records of their last 32-site tile crossed into the writer: the stub keeps 6/7 of the sites, n_alleles 2..4
    assert comm["records_sample_bytes_into_writer"] > 7 * 20 * 8 * 4 and comm["records_sample_GBps"] > 0
    assert line["ranks"]["evals_per_s_min"] <= line["ranks"]["evals_per_s_max"]
    # the short second leg: every tile's records gathered inside the timed step, reported beside `value` (which gathers none)
    leg = line["record_gather_leg"]
    assert leg["status"] == "ok" and leg["sites_per_rank"] == 64 and leg["bytes_into_writer"] > 0 and line["value_with_record_gather"] > 0
    assert "value_with_record_gather" not in early
    # value = the units ALL ranks processed / the slowest rank's time
    assert abs(line["value"] - 8 * 64 * 8 * 2 / (line["ms_per_step"] * 2e-3)) / line["value"] < 1e-3
    full = json.load(open(tmp_path / "d.json"))
    assert len(full["ranks"]["evals_per_s"]["per_rank"]) == 8 and len(full["ranks"]["kernel_ms_per_launch"]) == 8


def test_a_rank_that_never_arrives_in_the_record_gather_does_not_cost_the_line():
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--comm-timeout", "8", "--backend", "gloo"] + COMMON, env=_env(BENCH_TEST_STALL_RANK="2"),
                       capture_output=True, text=True, timeout=600)
    assert time.time() - t0 < 120
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:])       # `value` is complete: the run is not failed by a transfer reported beside it ...
    lines = _lines(r.stdout)
    assert len(lines) == 2 and lines[0]["n_gpus"] == 4 and lines[0]["value"] > 0     # ... the line is there, printed before the gather was attempted,
    assert lines[0]["comm"]["records_sample"] == "pending"
    assert lines[1]["comm"]["records_sample"] == "stalled" and lines[1]["value"] == lines[0]["value"]    # and once more, last, SAYING that the transfer stalled
    assert "stalled" in r.stderr


def test_under_a_launcher_the_same_line(tmp_path):
    # the driver's own command form: torch.distributed.run starts the ranks, bench.py joins them
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29533",
                        BENCH, "--gpus", "2", "--gather", "records", "--detail-file", str(tmp_path / "d.json")] + COMMON, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _lines(r.stdout)[-1]
    assert line["n_gpus"] == 2 and line["config"]["gather"] == "records" and line["records_gather"]["bytes_per_step_at_writer"] > 0
