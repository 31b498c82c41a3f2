#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (counter_collection.csv) -> profiles/pmc_traffic.json, the
per-launch HBM bytes bench.py reports as roofline.traffic.  Units and corrections as prescribed by
MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests
as 64 B, so it is doubled; WRITE_SIZE is taken as is.
usage: pmc_traffic.py <workload> <source-label> <dir-or-csv> [...]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

STAGES = {"k_inter": "k_inter", "k_residual": "k_residual", "k_intra": "k_intra<", "k_deblock": "k_deblock", "k_sao": "k_sao<",
          "k_meta": "k_meta", "k_intra_plan": "k_intra_plan", "k_tu_plan": "k_tu_plan", "k_job": "k_job_", "fill": "fillBuffer"}


def main(workload, label, paths):
    per = defaultdict(lambda: defaultdict(float))    # (kernel, counter) -> per-dispatch sums
    for path in paths:
        files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            for row in csv.DictReader(open(f)):
                per[(row["Kernel_Name"], row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    out = {"_source": label, "_units": "bytes per kernel launch; fetch = FETCH_SIZE KiB x 1024 x 2 (gfx950), write = WRITE_SIZE KiB x 1024"}
    for stage, pat in STAGES.items():
        fetch = [v for (k, c), d in per.items() if pat in k and c == "FETCH_SIZE" for v in d.values()]
        write = [v for (k, c), d in per.items() if pat in k and c == "WRITE_SIZE" for v in d.values()]
        if fetch and write:
            out[stage] = {"fetch_bytes": int(sum(fetch) / len(fetch) * 1024 * 2), "write_bytes": int(sum(write) / len(write) * 1024),
                          "launches_sampled": len(fetch)}
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    try:
        allw = json.load(open(dst))
    except Exception:
        allw = {}
    allw[workload] = out
    json.dump(allw, open(dst, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
