"""The line bench.py prints last is what the driver parses: it must stay below 4 KB, be strict JSON (no NaN / Infinity) and carry the
contract's keys with `roofline` and `cpu_baseline` (a 26.5 KB line was dropped and the round went unmeasured).
Built here, without a GPU, from a full result of the kind run_workload() returns (the committed round-4 line with every block)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "roofline", "cpu_baseline"]


def _opt(tmp_path, **kw):
    o = argparse.Namespace(steps=20, warmup=5, scaling="weak", tile_sites=65536, backend="nccl", detail_file=str(tmp_path / "detail.json"))
    o.__dict__.update(kw)
    return o


def _full_result():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r04_v9_bench.json")).read().strip().splitlines()[-1])
    res = {k: d[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "roofline", "ctx", "cpu_baseline", "record_packing")}
    res["workload"] = d["config"]["workload"]
    return res, d["extra"]


def _strict(text):
    def bad(c):
        raise ValueError(c)
    return json.loads(text, parse_constant=bad)


def test_headline_is_small_strict_and_complete(tmp_path, capsys):
    res, extra = _full_result()
    text = bench.emit(res, extra, _opt(tmp_path), 1, bench.METRIC)
    out = capsys.readouterr().out.strip().splitlines()
    assert out[-1] == text and len(out) == 2
    assert len(text) < bench.LINE_LIMIT, len(text)
    line = _strict(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["value"] == res["value"] and line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5
    assert line["dtype"].startswith("f64") and "f32" in line["dtype"]                 # float64 results; float32 only decides behind explicit bounds
    assert line["extra_oracle_pin"] == {k: "n<=3 goldens + model definition" for k in ("c2", "gl1q") if k in extra}
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "issue_frac", "issue_cycles_per_inst", "sclk_mhz", "mclk_mhz", "step_frac"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    assert all(not isinstance(v, (dict, list)) or k == "kernel_ms_per_launch" for k, v in rf.items())
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert set(line["extra"]) == set(extra) and all(isinstance(v, float) for v in line["extra"].values())
    # the extras' own line: short too, one object per workload
    ex = _strict(out[0])
    assert len(out[0]) < bench.LINE_LIMIT and set(ex["bench_extra"]) == set(extra)
    assert ex["bench_extra"]["fixedq"]["kernel"] == "k_gl" and ex["bench_extra"]["c5"]["value"] > 1e10
    # nothing is lost: the detail file holds the unabridged blocks
    full = json.load(open(tmp_path / "detail.json"))
    assert full["roofline"]["valu"]["frac"] > 0 and full["extra"]["precise"]["roofline"]["kernel_ms_total"]["k_gl"] > 0


def test_non_finite_numbers_never_reach_the_line(tmp_path, capsys):
    res, extra = _full_result()
    res["roofline"]["traffic"] = float("nan")
    res["roofline"]["issue"] = {"issue_frac": float("inf"), "issue_cycles_per_inst": 3.03, "issue_bound_ms": float("nan")}
    extra["c2"]["value"] = float("nan")
    text = bench.emit(res, extra, _opt(tmp_path), 1, bench.METRIC)
    line = _strict(text)
    assert line["roofline"]["traffic"] is None and line["roofline"]["issue_frac"] is None and line["roofline"]["issue_bound_ms"] is None


def test_issue_roof_and_clocks_reach_the_line(tmp_path, capsys):
    res, extra = _full_result()
    res["roofline"]["issue"] = {"issue_frac": 0.97, "issue_cycles_per_inst": 3.03, "issue_bound_ms": 7.6, "sclk_mhz_used": 2395.0}
    res["roofline"]["clocks"] = {"sclk_mhz": 2395.0, "sclk_mhz_min": 2100.0, "mclk_mhz": 2000.0, "samples": 170, "source": "/sys/bus/pci/devices/0000:05:00.0"}
    line = _strict(bench.emit(res, extra, _opt(tmp_path), 1, bench.METRIC))
    rf = line["roofline"]
    assert rf["issue_frac"] == 0.97 and rf["issue_cycles_per_inst"] == 3.03 and rf["sclk_mhz"] == 2395.0 and rf["mclk_mhz"] == 2000.0 and rf["sclk_mhz_min"] == 2100.0
    assert "valu_frac" not in rf


def test_multi_rank_blocks_fit_too(tmp_path, capsys):
    res, _ = _full_result()
    res.pop("cpu_baseline")
    res["ranks"] = {"evals_per_s": {"min": 4.1e9, "max": 4.3e9, "per_rank": [4.2e9] * 8}, "ms_per_step": {"min": 230.0, "max": 240.0},
                    "kernel_ms_per_launch": [{k: 1.0 for k in bench.KERNELS}] * 8, "note": "x" * 300}
    res["comm"] = {"backend": "nccl", "world": 8, "gather": "sample", "gather_ms": 1.25, "gather_note": "y" * 300, "records_sample": "ok",
                   "records_sample_ms": 40.0, "records_sample_bytes_into_writer": 29e9, "records_sample_GBps": 700.0,
                   "full_record_gather_s_per_step_at_that_rate": 0.65, "records_sample_note": "z" * 400}
    info = {"rccl_world": 8, "backend": "rccl (torch.distributed nccl)", "rccl_version": "2.26.6", "gather": "sample"}
    res["value_with_record_gather"] = 1.4e10
    res["record_gather_leg"] = {"status": "ok", "sites_per_rank": 262144, "ms_per_step": 150.0, "GBps_into_writer": 640.0, "bytes_into_writer": 9.6e10, "note": "n" * 300}
    text = bench.emit(res, {}, _opt(tmp_path), 8, bench.METRIC, info)
    line = _strict(text)
    assert len(text) < bench.LINE_LIMIT
    assert line["value_with_record_gather"] == 1.4e10 and line["record_gather_leg"]["status"] == "ok" and line["record_gather_leg"]["GBps_into_writer"] == 640.0
    assert line["n_gpus"] == 8 and line["config"]["rccl_world"] == 8 and line["config"]["rccl_version"] == "2.26.6"
    assert line["comm"]["records_sample_GBps"] == 700.0 and "records_sample_note" not in line["comm"]
    assert line["ranks"]["evals_per_s_min"] == 4.1e9
