#!/usr/bin/env python3
"""Per-kernel durations of the TIMED steps of a rocprofv3 kernel trace.
rocprofv3's own kernel_stats.csv averages every dispatch of the process, the untimed warm-up step included (whose first
launches pay for code loading and first-touch of the output planes: one 18 ms `k_gl` launch among 64 of 2.3 ms at fixed-q);
bench.py's `roofline.kernel_ms_total / launches` covers the timed steps only.  This reads the per-dispatch trace, drops the
first WARMUP/(WARMUP+STEPS) of each kernel's dispatches (bench.py launches the same kernels in every step) and writes what
remains in the layout of kernel_stats.csv, with the all-dispatch average beside it.
usage: tools/trace_timed.py <kernel_trace.csv> <warmup steps> <timed steps> > kernel_timed_stats.csv"""
import csv, statistics, sys
path, warm, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
by = {}
for r in csv.DictReader(open(path)):
    by.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
w.writerow(["Name", "Calls", "AverageNs", "TimedCalls", "TimedAverageNs", "TimedMedianNs", "TimedMinNs", "TimedMaxNs", "WarmupAverageNs"])
rows = []
for name, d in by.items():
    d.sort()
    dur = [x[1] for x in d]
    skip = len(dur) * warm // (warm + steps) if len(dur) % (warm + steps) == 0 else 0     # a kernel not launched per step: everything counts
    t = dur[skip:]
    rows.append([name, len(dur), sum(dur) / len(dur), len(t), sum(t) / len(t), statistics.median(t), min(t), max(t), (sum(dur[:skip]) / skip) if skip else 0.0])
for r in sorted(rows, key=lambda r: -r[3] * r[4]):
    w.writerow(r)
