#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel trace, separate PMC passes (FETCH_SIZE / WRITE_SIZE).
# usage: tools/gpu_round.sh <tag> [notests]      outputs under gpurun_out/<tag>/
TAG=${1:-run}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json
REPO=$PWD
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python $REPO/bench.py --no-cpu-baseline --pipeline-depth 1 > $OUT/kt.log 2>&1
python $REPO/tools/rocprof_summary.py $OUT/kt $OUT/kernel_stats.txt | head -30
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_rd -o rd --output-format csv -- python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 1 --pipeline-depth 1 > $OUT/pmc_rd.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_wr -o wr --output-format csv -- python $REPO/bench.py --no-cpu-baseline --steps 5 --warmup 1 --pipeline-depth 1 > $OUT/pmc_wr.log 2>&1
python $REPO/tools/pmc_summary.py $OUT/pmc_rd $OUT/pmc_wr > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
python $REPO/tools/pmc_traffic.py c5_8k10_8tiles "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of round $TAG (profiles/${TAG}_c5_pmc_summary.txt)" $OUT/pmc_rd $OUT/pmc_wr > /dev/null; cp $REPO/profiles/pmc_traffic.json $OUT/pmc_traffic.json
cd $REPO
# the N>1 code path of bench.py (process group, tile-sharded leg over RCCL) with the one GPU of this box
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --force-tile-shard --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; tail -c 1500 $OUT/bench_dist1.json; tail -3 $OUT/bench_dist1.err
# keep the merge-back small
find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*counter_collection.csv" -size +20M -delete
