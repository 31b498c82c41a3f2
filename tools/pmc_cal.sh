#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/ubench/ub_pmc_cal.hip) -> gpurun_out/pmc_cal/summary.txt
OUT=$PWD/gpurun_out/pmc_cal; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $c -d $OUT/$n -o x --output-format csv -- $REPO/tools/ubench/ub_pmc_cal > $OUT/$n.log 2>&1
done
python - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-28s %-24s %s" % (k, c, " ".join("%.0f" % x for x in v)))
PY
head -3 $OUT/FETCH_SIZE.log; cat $OUT/summary.txt
