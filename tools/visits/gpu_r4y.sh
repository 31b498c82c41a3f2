#!/bin/bash
# round 4, visit y: tiled-reference experiment, other tile shapes (64x8, 32x16, 128x4) beside 32x8 and the linear layout — one call
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/r4y; mkdir -p $O
export TMPDIR=/tmp
w=c5_8k10_8tiles
for v in base tiled tiled64x8 tiled32x16 tiled128x4; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  if [ "$v" != base ]; then timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/tests_$v.log 2>&1; echo "$v parity rc=$? $(tail -1 $O/tests_$v.log)" | tee -a $O/summary.txt; fi
  cd $REPO
  timeout 300 python bench.py --no-cpu-baseline --no-with-upload --no-end-to-end --no-dependent-chain --workload $w --steps 200 --warmup 10 --pipeline-depth 3 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s depth 3: %.4f ms/pic  one-at-a-time %.4f  inter=%.4f' % ('$v', d['ms_per_step'], d['ms_per_step_one_in_flight'], d['stage_ms']['inter']))" | tee -a $O/summary.txt
  cd /tmp; rm -rf /tmp/rd_$v
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/rd_$v -o rd --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1 > $O/pmc_rd_$v.log 2>&1
  python $REPO/tools/pmc_summary.py /tmp/rd_$v > $O/pmc_summary_$v.txt 2>&1; grep -i "k_inter_jobs" $O/pmc_summary_$v.txt | head -2 | cut -c1-200 | sed "s/^/$v  /" >> $O/summary.txt
done
cat $O/summary.txt
