#!/bin/bash
# round 4, visit i: end-to-end attribution; kernel trace + PMC passes of the product configuration (read-modify-write residuals) for C5 and C2
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 900 bash tools/e2e_attrib.sh $OUT/e2e_attribution.txt > /dev/null 2>&1; cat $OUT/e2e_attribution.txt
cd /tmp
LIGHT="--steps 60 --warmup 5 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1"
for w in c5_8k10_8tiles c2_1080p_intra c3_4k_inter c4_4k_4tiles; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$w -o kt -- python $REPO/bench.py --workload $w $LIGHT > $OUT/bench_${w}_kt.json 2> $OUT/kt_$w.log
  python $REPO/tools/rocprof_summary.py $OUT/kt_$w $OUT/kernel_stats_$w.txt | head -12
  timeout 400 python $REPO/bench.py --workload $w --steps 100 --warmup 10 --no-end-to-end > $OUT/bench_$w.json 2>> $OUT/bench.err
done
for w in c5_8k10_8tiles c2_1080p_intra; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_rd_$w -o rd --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1 > $OUT/pmc_rd_$w.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_wr_$w -o wr --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1 > $OUT/pmc_wr_$w.log 2>&1
  python $REPO/tools/pmc_summary.py $OUT/pmc_rd_$w $OUT/pmc_wr_$w > $OUT/pmc_summary_$w.txt 2>&1
  python $REPO/tools/pmc_traffic.py $w "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of round 4 (profiles/r04_i_${w}_pmc_summary.txt)" $OUT/pmc_rd_$w $OUT/pmc_wr_$w > /dev/null
done
cp $REPO/profiles/pmc_traffic.json $OUT/pmc_traffic.json
cat $OUT/pmc_summary_c5_8k10_8tiles.txt
cd $REPO
python -c "
import json
for w in ('c2_1080p_intra','c3_4k_inter','c4_4k_4tiles','c5_8k10_8tiles'):
    d=json.loads(open('$OUT/bench_%s.json' % w).read().strip().splitlines()[-1]); print(w, d['value'], d['ms_per_step'], d['ms_per_step_one_in_flight'], d['stage_ms'], d.get('with_upload',{}).get('submit_only'), d['cpu_baseline']['value'])"
find $OUT -name "*.db" -size +10M -delete; find $OUT -name "*counter_collection.csv" -size +10M -delete; find $OUT -name "*kernel_trace.csv" -size +10M -delete
