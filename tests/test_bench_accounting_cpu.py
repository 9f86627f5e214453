"""bench.py's accounting without a GPU: the algorithmic bytes per evaluation of every workload follow SURVEY section 8d's formula, every
workload the bench times has committed counters (profiles/pmc_traffic.json) with the fields the roofline block quotes, and the pool-loop
cost account (profiles/r04_loop_cost.json, tools/loop_cost.py) is self-consistent."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench


def test_algorithmic_bytes_follow_the_survey_formula():
    # B_eval = 1 (packed GT in) + 4 (DP) + 4 G (GL) [+ 4 G (PL), or G as pl_u8] [+ 4 G (GP)] [+ 16 per AD-type FORMAT tag]
    assert bench.algorithmic_bytes_per_eval(["fmt_dp", "gl"], 15) == 65
    assert bench.algorithmic_bytes_per_eval(["fmt_dp", "gl"], 10) == 45
    assert bench.algorithmic_bytes_per_eval(["fmt_dp", "gl", "pl"], 15) == 125
    assert bench.algorithmic_bytes_per_eval(["fmt_dp", "gl", "pl_u8"], 15) == 80
    assert bench.algorithmic_bytes_per_eval(bench.WORKLOADS["alltags"]["fields"], 15) == 1 + 4 + 3 * 60 + 3 * 16
    assert bench.algorithmic_bytes_per_eval(bench.WORKLOADS["qsi16"]["fields"], 15) == 65       # INFO/QS, INFO/I16 are per site


def test_every_timed_workload_has_committed_counters():
    prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for wl in bench.WORKLOADS:
        assert wl in prof, wl
        e = prof[wl]
        assert e["source"].startswith("r06_") and len(e["src_sha"]) == 16
        k = e["kernels"]
        assert "k_gl" in k or {"k_gl2", "k_gl2_scan", "k_gl_redo"} <= set(k), wl          # (k_gl2: fixed-q, depth 30 -- DESIGN.md section 4.5b)
        for name, v in k.items():
            if name == "k_sample_seg_list":                      # the segment loop's follow-up kernel: nothing listed at the bench configurations, its wavefronts leave at once
                continue
            assert v["sites_per_launch"] > 0 and v["valu_insts_per_wave"] > 0 and 0 < v["active_lanes_per_valu_inst"] < 70, (wl, name)
            assert v["hbm_bytes_per_launch"] > 0 and 0 <= v["valu_busy_frac"] <= 1.0, (wl, name)       # (clamped at 1: SQ_ACTIVE_INST_VALU counts 4 cycles per instruction)
    assert "k_siteagg" in prof["alltags"]["kernels"] and "k_siteagg" in prof["qsi16"]["kernels"] and "k_redo" in prof["c3"]["kernels"]
    for wl in bench.WORKLOADS:                          # the per-workload summaries and kernel traces the line points at exist
        tag = prof[wl]["source"]
        assert os.path.exists(os.path.join(ROOT, "profiles", tag + "_pmc_summary.json")), tag
        assert os.path.exists(os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")), tag


def test_issue_roof_is_a_lower_bound_of_the_committed_kernel_times():
    """profiles/issue_roof.json: for every workload it covers, the bound bench.py computes -- pool iterations x (replayed loop path +
    rare blocks) + the instructions outside the loop at the part's cheapest rate -- lies at or below the kernel time of the committed trace, and above
    0.9 of it (the kernels it covers are issue-bound); the entry belongs to the build the counters were taken from."""
    import csv
    ir = json.load(open(os.path.join(ROOT, "profiles", "issue_roof.json")))
    prof = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert {"c3", "c4", "gl1q", "qsi16", "alltags"} <= set(ir)
    for wl, e in ir.items():
        lp = e["loop"]
        assert e["kernel"] == "k_sample" and e["src_sha"] == prof[wl]["src_sha"], wl
        assert 2.9 < e["issue_cycles_per_inst"] < 3.4 and lp["floor_cycles_per_inst_outside_the_loop"] < 2.5
        assert lp["simd_cycles_per_iteration_main_path"] < lp["class_sum_cycles_main_path"]          # the classes do not add: the replay is cheaper than their sum
        vpw = prof[wl]["kernels"]["k_sample"]["valu_insts_per_wave"]
        it = lp["iterations_per_wave"]
        loop_valu = it * (lp["valu_per_iteration_main_path"] + lp["rare_blocks_valu_per_iteration"])
        assert 0.7 * vpw < loop_valu < vpw, wl
        cyc = it * (lp["simd_cycles_per_iteration_main_path"] + lp["rare_blocks_cycles_per_iteration"]) + (vpw - loop_valu) * lp["floor_cycles_per_inst_outside_the_loop"]
        n = bench.WORKLOADS[wl]["samples"]
        waves = 65536 * ((n + 63) // 64)
        rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", prof[wl]["source"] + "_kernel_timed_stats.csv"))))
        ms = max(float(r["TimedMedianNs"]) for r in rows if "k_sample_seg" in r["Name"] and ", 1, " in r["Name"]) * 1e-6      # a full 65536-site launch
        frac = waves * cyc / (bench.N_SIMD * bench.SCLK_NOMINAL_MHZ * 1e6) * 1e3 / ms
        assert 0.88 < frac <= 1.0, (wl, frac, ms)


def test_pool_loop_cost_account_is_consistent():
    lc = json.load(open(os.path.join(ROOT, "profiles", "r04_loop_cost.json")))
    roles = {b["role"] for b in lc["blocks"]}
    assert {"common", "finish", "slow_n", "slow_g"} <= roles
    common = sum(b["valu"] for b in lc["blocks"] if b["role"] == "common")
    assert common == lc["valu_common_path"] and 50 <= common <= 70
    for b in lc["blocks"]:
        assert abs(b["cycles"] - sum(lc["cost_table_cycles"][k] * n for k, n in b["classes"].items())) < 1e-6
        assert sum(b["classes"].values()) == b["valu"]
    assert 3.5 < lc["avg_cycles_per_valu_inst"] < 5.0
    assert abs(lc["simd_cycles_per_iteration_weighted"] / lc["valu_per_iteration_weighted"] - lc["avg_cycles_per_valu_inst"]) < 0.01
