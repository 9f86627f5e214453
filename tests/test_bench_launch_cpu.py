"""bench.py's launch logic without a GPU: `--gpus N` must start N ranks itself or, under a launcher, refuse a world size
that differs from --gpus -- never silently report n_gpus: 1."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _have_gpu():
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return False


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e.update(kw)
    return e


def test_world_size_mismatch_is_refused():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="3"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE is 3" in r.stderr
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "--gpus 1 but WORLD_SIZE is 2" in r.stderr


@pytest.mark.skipif(_have_gpu(), reason="needs a machine WITHOUT a GPU")
def test_gpus_n_spawns_n_ranks_which_fail_loudly_without_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--no-cpu-baseline", "--no-extra"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs GPU" in r.stderr and "0 device(s)" in r.stderr        # a rank per requested GPU was started and said what it lacks
    assert '"n_gpus"' not in r.stdout                                   # and no result line was fabricated


def test_one_failing_rank_ends_the_run_at_once():
    """a rank that exits early (missing GPU, context or RCCL init failure) while the others are blocked -- in a rendezvous, a
    collective -- must end the launcher with its error right away, not after the process group's timeout"""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3"], env=_env(BENCH_TEST_LAUNCHER="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "rank 1 fails" in r.stderr
    assert time.time() - t0 < 30
    assert '"n_gpus"' not in r.stdout
