#!/usr/bin/env python3
"""Steady-state per-kernel statistics from a rocprofv3 --kernel-trace run (CSV output).

  kernel_stats.py <rocprof output dir> [--skip W] [--match sg_]

rocprofv3's own --stats averages every dispatch, including the warm-up launches (cold instruction cache, first touch
of the posting store): round 1's committed average (2.650 ms) sat above the steady state the driver timed (2.468 ms).
This summary drops the first W dispatches of every matching kernel (W = bench.py's --warmup) and prints the same
columns as rocprofv3's *_kernel_stats.csv, so the committed file is the steady state the bench's HIP events measure.
"""
import argparse
import csv
import glob
import os
import statistics
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--skip", type=int, default=0, help="warm-up dispatches to drop per matching kernel")
    ap.add_argument("--match", default="sg_", help="kernels the skip applies to (substring)")
    args = ap.parse_args()
    files = sorted(glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True))
    if not files:
        sys.exit("no *kernel_trace.csv under %s" % args.dir)
    per = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            per.setdefault(r["Kernel_Name"], []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    rows = []
    for name, d in per.items():
        d.sort()
        dur = [x[1] for x in d]
        skipped = 0
        if args.match in name and len(dur) > args.skip:
            skipped = args.skip
            dur = dur[args.skip:]
        rows.append((name, len(dur), sum(dur), skipped, dur))
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev", "WarmupDispatchesDropped"])
    for name, n, tot, skipped, dur in sorted(rows, key=lambda r: -r[2]):
        w.writerow([name, n, tot, round(tot / n, 1), round(100.0 * tot / total, 4), min(dur), max(dur),
                    round(statistics.pstdev(dur), 1) if n > 1 else 0.0, skipped])


if __name__ == "__main__":
    main()
