"""tools/trace_timed.py: the timed-step statistics of a kernel trace drop exactly the warm-up share of each kernel's dispatches."""
import csv
import io
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_timed_steps_only(tmp_path):
    p = tmp_path / "1_kernel_trace.csv"
    with open(p, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        t = 0
        for step in range(4):                               # 1 warm-up + 3 timed steps, two launches of k_a per step
            for j in range(2):
                d = 9000 if step == 0 else 1000 + 10 * j    # the warm-up launches are 9x longer
                w.writerow(["KERNEL_DISPATCH", "k_a(P, T)", t, t + d]); t += d + 5
            if step == 0:
                w.writerow(["KERNEL_DISPATCH", "k_once()", t, t + 77]); t += 80   # not launched per step: everything counts
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_timed.py"), str(p), "1", "3"], capture_output=True, text=True, check=True)
    rows = {x["Name"]: x for x in csv.DictReader(io.StringIO(r.stdout))}
    a = rows["k_a(P, T)"]
    assert int(a["Calls"]) == 8 and int(a["TimedCalls"]) == 6
    assert float(a["TimedAverageNs"]) == 1005.0 and float(a["WarmupAverageNs"]) == 9000.0 and float(a["TimedMaxNs"]) == 1010
    assert abs(float(a["AverageNs"]) - (2 * 9000 + 3 * 2010) / 8) < 1e-9
    o = rows["k_once()"]
    assert int(o["TimedCalls"]) == 1 and float(o["TimedAverageNs"]) == 77.0
