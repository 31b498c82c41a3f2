#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean counter value per dispatch.
usage: pmc_summary.py <dir-or-csv> [...]   (separate --pmc passes may be given together)"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    for path in paths:
        files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            per_dispatch = defaultdict(float)
            for row in csv.DictReader(open(f)):
                per_dispatch[(row["Kernel_Name"], row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
            for (k, _d, c), v in per_dispatch.items():
                acc[k][c].append(v)
    counters = sorted({c for k in acc for c in acc[k]})
    print("%-60s %6s " % ("kernel", "calls") + " ".join("%16s" % c[-16:] for c in counters))
    for k in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_WAVE_CYCLES", [0]))):
        n = max(len(v) for v in acc[k].values())
        print("%-60s %6d " % (k[:60], n) + " ".join("%16.4g" % (sum(acc[k][c]) / len(acc[k][c])) if c in acc[k] else "%16s" % "-" for c in counters))


if __name__ == "__main__":
    main(sys.argv[1:])
