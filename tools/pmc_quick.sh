#!/bin/bash
OUT=$PWD/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT; REPO=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_rd -o rd --output-format csv -- python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 1 --pipeline-depth 1 > $OUT/pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_wr -o wr --output-format csv -- python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 1 --pipeline-depth 1 > $OUT/pmc_wr.log 2>&1
timeout 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_c -o c --output-format csv -- python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 1 --pipeline-depth 1 > $OUT/pmc_c.log 2>&1
python $REPO/tools/pmc_summary.py $OUT/pmc_rd $OUT/pmc_wr $OUT/pmc_c 2>&1 | grep -E "kernel|k_inter"
find $OUT -name "*.csv" -size +5M -delete
