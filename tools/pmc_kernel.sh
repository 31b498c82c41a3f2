#!/bin/bash
# Counters of one kernel of the bench picture.  usage: tools/pmc_kernel.sh <kernel substring> <workload> "<counters of pass 1>" ["<pass 2>" ...]
K=$1; W=$2; shift 2
REPO=$PWD; export TMPDIR=/tmp
OUT=$REPO/gpurun_out/pmc_kernel; rm -rf $OUT; mkdir -p $OUT; cd /tmp
i=0
for c in "$@"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $c -d $OUT/p$i -o x --output-format csv -- python $REPO/bench.py --workload $W --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --steps 4 --warmup 1 --pipeline-depth 1 > $OUT/p$i.log 2>&1 || tail -2 $OUT/p$i.log
done
python $REPO/tools/pmc_summary.py $OUT 2>&1 | grep -E "^kernel|$K" | tee $REPO/gpurun_out/pmc_kernel_$K_$W.txt
rm -rf $OUT
