#!/usr/bin/env python3
"""Concurrency of the pipelined decode from a rocprofv3 --kernel-trace CSV: per kernel the mean duration under overlap, the union
of all kernel intervals (GPU busy with at least one kernel), the sum of durations, and the mean number of kernels in flight.
usage: overlap_stats.py <dir with *kernel_trace.csv> <pictures in the timed region> [skip_fraction=0.3]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(path, n_pic, skip=0.3):
    f = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in csv.DictReader(open(f))]
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + int((t1 - t0) * skip)                   # skip set-up / warm-up / the one-at-a-time leg at the start
    rows = [r for r in rows if r[0] >= lo]
    wall = max(r[1] for r in rows) - rows[0][0]
    total = sum(e - s for s, e, _ in rows)
    union, cur_s, cur_e = 0, None, None
    for s, e, _ in rows:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    per = defaultdict(list)
    for s, e, k in rows:
        per[k].append(e - s)
    print("window %.3f ms, %d kernels; GPU busy (union of kernel intervals) %.1f %%; sum of kernel durations / window = %.2f kernels in flight on average"
          % (wall / 1e6, len(rows), 100.0 * union / wall, total / wall))
    print("%-60s %8s %10s" % ("kernel", "calls", "avg us"))
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print("%-60s %8d %10.2f" % (k[:60], len(v), sum(v) / len(v) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), float(sys.argv[3]) if len(sys.argv) > 3 else 0.3)
