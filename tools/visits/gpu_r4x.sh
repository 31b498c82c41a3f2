#!/bin/bash
# round 4, visit x: THE TILED-REFERENCE EXPERIMENT.  libde265_amd/variants/tiled.so (tools/variants.sh tiled
# "-DM355_X_TILED"): k_inter_jobs' FAST path reads a tiled copy (32 x 8 samples + apron) of every reference frame, made by one
# conversion kernel per plane.  One call: parity of the variant, A/B of the bench (depth 1 and 3), kernel traces, PMC FETCH / WRITE.
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/r4x; mkdir -p $O
export TMPDIR=/tmp
V=$REPO/libde265_amd/variants/tiled.so
M355_LIB=$V timeout 900 python -m pytest tests/test_gpu_synth.py tests/test_gpu_pipeline.py tests/test_gpu_random.py -x -q -m gpu > $O/tests_tiled.log 2>&1; echo "tiled variant parity rc=$?" | tee -a $O/summary.txt
tail -2 $O/tests_tiled.log | tee -a $O/summary.txt
B="--no-cpu-baseline --no-with-upload --no-end-to-end"
for rep in 1 2; do for v in base tiled; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$V; fi
  for w in c5_8k10_8tiles c3_4k_inter; do
    timeout 300 python bench.py $B --workload $w --steps 200 --warmup 10 --pipeline-depth 3 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s %-16s depth 3: %.4f ms/pic  one-at-a-time %.4f  dependent chain %.4f  %s' % ('$v', '$w', d['ms_per_step'], d['ms_per_step_one_in_flight'], d['dependent_chain']['ms_per_step'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $O/summary.txt
  done
done; done
cd /tmp
w=c5_8k10_8tiles
for v in base tiled; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$V; fi
  rm -rf /tmp/kt_$v; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o kt -- python $REPO/bench.py --workload $w --steps 30 --warmup 3 --pipeline-depth 1 --no-cpu-baseline --no-with-upload --no-end-to-end > /dev/null 2>$O/kt_$v.err
  echo "--- kernel trace, $v (depth 1; the dependent-chain leg converts two references per picture)" >> $O/summary.txt
  python $REPO/tools/rocprof_summary.py /tmp/kt_$v $O/kernel_stats_$v.txt | head -16 | cut -c1-170 >> $O/summary.txt
  rm -rf /tmp/rd_$v /tmp/wr_$v
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d /tmp/rd_$v -o rd --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1 > $O/pmc_rd_$v.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d /tmp/wr_$v -o wr --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1 > $O/pmc_wr_$v.log 2>&1
  echo "--- PMC (FETCH_SIZE x 2 on gfx950, WRITE_SIZE), $v" >> $O/summary.txt
  python $REPO/tools/pmc_summary.py /tmp/rd_$v /tmp/wr_$v > $O/pmc_summary_$v.txt 2>&1; grep -i "k_inter\|k_tile\|kernel" $O/pmc_summary_$v.txt | head -8 | cut -c1-200 >> $O/summary.txt
done
cat $O/summary.txt
