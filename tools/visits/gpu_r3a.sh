#!/bin/bash
# round 3, visit a: the rewritten k_intra (plan kernel + lean chain) against the previous library, on hardware.
OUT=$PWD/gpurun_out/r03a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
for v in old base nw8; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$PWD/libde265_amd/variants/$v.so; fi
  for w in c2_1080p_intra c3_4k_inter c5_8k10_8tiles; do
    for d in 1 3 8; do
      [ $w != c2_1080p_intra ] && [ $d = 8 ] && continue
      timeout 300 python bench.py $B --workload $w --steps 200 --warmup 10 --pipeline-depth $d 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-5s %-16s depth %d: %.4f ms/pic  one-at-a-time %.4f  %s' % ('$v', '$w', $d, d['ms_per_step'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $OUT/variants.txt
    done
  done
  timeout 300 python tools/diag_intra.py 1 4 16 2>>$OUT/bench.err | sed "s/^/$v /" | tee -a $OUT/diag_intra.txt
done
unset M355_LIB
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_c2 -o kt -- python $OLDPWD/bench.py --workload c2_1080p_intra $B --steps 30 --warmup 3 --pipeline-depth 1 > $OUT/kt_c2.json 2> $OUT/kt_c2.log
cd $OLDPWD; python tools/rocprof_summary.py $OUT/kt_c2 $OUT/kernel_stats_c2.txt | head -14
find $OUT -name "*.db" -size +5M -delete; find $OUT -name "*kernel_trace.csv" -size +5M -delete
