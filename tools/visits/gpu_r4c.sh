#!/bin/bash
# round 4, visit c: k_sao / k_deblock with their loads in two round trips vs the previous build (libde265_amd/variants/r04_old_filters.so), ONE box
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
for v in ${VARS:-base r04_old_filters base r04_old_filters}; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$PWD/libde265_amd/variants/$v.so; fi
  for w in ${WORKLOADS:-c5_8k10_8tiles c3_4k_inter c2_1080p_intra}; do
    for d in 3 1; do
      timeout 300 python bench.py $B --workload $w --steps ${STEPS:-200} --warmup 10 --pipeline-depth $d 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-16s %-16s depth %d: %.4f ms/pic  one-at-a-time %.4f  %s' % ('$v', '$w', $d, d['ms_per_step'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $OUT/variants.txt
    done
  done
done
unset M355_LIB
cd /tmp; rocprofv3 --kernel-trace --stats -d $OUT/prof -o trace -- python $OLDPWD/bench.py $B --steps 60 --warmup 5 --pipeline-depth 1 > $OUT/prof.log 2>&1; cd $OLDPWD
python tools/rocprof_summary.py $OUT/prof $OUT/kernel_stats.txt > /dev/null 2>&1; head -24 $OUT/kernel_stats.txt
find $OUT -name "*.db" -size +10M -delete
