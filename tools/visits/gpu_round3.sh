#!/bin/bash
# One GPU-box visit of round 3: parity tests, the driver's bench line, rocprofv3 kernel traces of every BASELINE config (one
# picture in flight), separate PMC passes (FETCH_SIZE / WRITE_SIZE) for C5 and C2, the N>1 path at world size 1, end-to-end.
# usage: tools/gpu_round2.sh <tag> [notests]      outputs under gpurun_out/<tag>/
TAG=${1:-r03}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
REPO=$PWD
if [ "$2" != "notests" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
  tail -3 $OUT/pytest.log
fi
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real   # what the driver runs
tail -c 5000 $OUT/bench.json
LIGHT="--steps 60 --warmup 5 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1"
cd /tmp
for w in c2_1080p_intra c3_4k_inter c4_4k_4tiles c5_8k10_8tiles; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$w -o kt -- python $REPO/bench.py --workload $w $LIGHT > $OUT/bench_${w}_kt.json 2> $OUT/kt_$w.log
  python $REPO/tools/rocprof_summary.py $OUT/kt_$w $OUT/kernel_stats_$w.txt | head -14
  timeout 400 python $REPO/bench.py --workload $w --steps 100 --warmup 10 --no-end-to-end > $OUT/bench_$w.json 2>> $OUT/bench.err   # with cpu_baseline, 25 timed regions
done
for w in c5_8k10_8tiles c2_1080p_intra; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_rd_$w -o rd --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1 > $OUT/pmc_rd_$w.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_wr_$w -o wr --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1 > $OUT/pmc_wr_$w.log 2>&1
  python $REPO/tools/pmc_summary.py $OUT/pmc_rd_$w $OUT/pmc_wr_$w > $OUT/pmc_summary_$w.txt 2>&1
  python $REPO/tools/pmc_traffic.py $w "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of round $TAG (profiles/${TAG}_${w}_pmc_summary.txt)" $OUT/pmc_rd_$w $OUT/pmc_wr_$w > /dev/null
done
cp $REPO/profiles/pmc_traffic.json $OUT/pmc_traffic.json
cd $REPO
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 --force-tile-shard --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err; tail -c 900 $OUT/bench_dist1.json; tail -2 $OUT/bench_dist1.err
# all-intra pictures in flight: the default (4 hardware queues per stream priority) and the tuned setting for intra-only streams
for cfg in "4 3" "4 9" "16 8"; do set -- $cfg; GPU_MAX_HW_QUEUES=$1 timeout 300 python bench.py --workload c2_1080p_intra --steps 200 --warmup 10 --pipeline-depth $2 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 GPU_MAX_HW_QUEUES=$1 depth $2: %.4f ms/pic = %.3f M CTB64/s (one at a time %.4f)' % (d['ms_per_step'], 510/d['ms_per_step']/1e3, d['ms_per_step_one_in_flight']))" | tee -a $OUT/c2_in_flight.txt; done
timeout 300 python tools/diag_intra.py 1 3 2>>$OUT/bench.err | tee $OUT/diag_intra.txt
bash tools/e2e.sh $OUT > /dev/null 2>&1; grep -A1 "t 8\|md5" $OUT/e2e.txt | grep -v "^--"
find $OUT -name "*.db" -size +10M -delete; find $OUT -name "*counter_collection.csv" -size +10M -delete; find $OUT -name "*kernel_trace.csv" -size +10M -delete
