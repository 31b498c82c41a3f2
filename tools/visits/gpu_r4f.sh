#!/bin/bash
# round 4, visit f: tests, the driver's line, host phases of the submit, kernel trace + PMC traffic of C5
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_spread'] if 'ms_per_step_spread' in d else '', d['stage_ms'], d['roofline']['frac'], d['with_upload']['submit_only'], d['with_upload']['ms_per_step'], d['end_to_end'].get('speedup'), d['end_to_end']['with_output'].get('speedup'))"
M355_PROFILE_UPLOAD=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dependent-chain --no-end-to-end > $OUT/bench_upload.json 2> $OUT/upload_profile.txt
grep "in place" $OUT/upload_profile.txt | tail -4; grep "m355 submit" $OUT/upload_profile.txt | sed -n 60,63p
LIGHT="--steps 60 --warmup 5 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1"
cd /tmp
w=c5_8k10_8tiles
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$w -o kt -- python $REPO/bench.py --workload $w $LIGHT > $OUT/bench_${w}_kt.json 2> $OUT/kt_$w.log
python $REPO/tools/rocprof_summary.py $OUT/kt_$w $OUT/kernel_stats_$w.txt | head -22
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_rd_$w -o rd --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1 > $OUT/pmc_rd_$w.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_wr_$w -o wr --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1 > $OUT/pmc_wr_$w.log 2>&1
python $REPO/tools/pmc_summary.py $OUT/pmc_rd_$w $OUT/pmc_wr_$w > $OUT/pmc_summary_$w.txt 2>&1; cat $OUT/pmc_summary_$w.txt | head -40
python $REPO/tools/pmc_traffic.py $w "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of round $TAG (profiles/${TAG}_${w}_pmc_summary.txt)" $OUT/pmc_rd_$w $OUT/pmc_wr_$w > /dev/null
cp $REPO/profiles/pmc_traffic.json $OUT/pmc_traffic.json
cd $REPO
find $OUT -name "*.db" -size +10M -delete; find $OUT -name "*counter_collection.csv" -size +10M -delete; find $OUT -name "*kernel_trace.csv" -size +10M -delete
