#!/bin/bash
# round 4, visit k: k_deblock with the sample loads behind the edge flags vs unconditional (one box), glue ranks after the parallel split, driver line
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -2 $OUT/pytest.log
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
for v in base r04_deblock_uncond base r04_deblock_uncond; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$PWD/libde265_amd/variants/$v.so; fi
  for w in c5_8k10_8tiles c3_4k_inter; do
    timeout 300 python bench.py $B --workload $w --steps 200 --warmup 10 --pipeline-depth 3 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-20s %-16s depth 3: %.4f ms/pic  one-at-a-time %.4f  %s' % ('$v', '$w', d['ms_per_step'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $OUT/variants.txt
  done
done
unset M355_LIB
oracle/_ref/streamgen /tmp/a8k.h265 7680 4320 10 4 2 16 77 5 1 1 >/dev/null 2>&1
for n in 1 2 4 8; do echo "M355_GLUE_RANKS=$n (8K 10-bit 4x2 tiles, dec265 -q -t 8): $(M355_GLUE_RANKS=$n M355_PIPELINE_DEPTH=3 glue/_build/dec265 -q -t 8 /tmp/a8k.h265 2>&1 | grep -o '@ *[0-9.]* fps')"; done | tee $OUT/glue_ranks_one_gpu.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_spread'], d['stage_ms'], d['roofline']['frac'], d['with_upload']['submit_only'], d['with_upload']['ms_per_step'], d['end_to_end'].get('speedup'), d['end_to_end']['with_output'].get('speedup'))"
cd /tmp; w=c5_8k10_8tiles
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_rd_$w -o rd --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1 > $OUT/pmc_rd_$w.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_wr_$w -o wr --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --pipeline-depth 1 > $OUT/pmc_wr_$w.log 2>&1
python $REPO/tools/pmc_summary.py $OUT/pmc_rd_$w $OUT/pmc_wr_$w > $OUT/pmc_summary_$w.txt 2>&1; grep "deblock\|sao" $OUT/pmc_summary_$w.txt
python $REPO/tools/pmc_traffic.py $w "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of round 4 (profiles/r04_k_${w}_pmc_summary.txt)" $OUT/pmc_rd_$w $OUT/pmc_wr_$w > /dev/null
cp $REPO/profiles/pmc_traffic.json $OUT/pmc_traffic.json
cd $REPO
find $OUT -name "*.db" -size +10M -delete; find $OUT -name "*counter_collection.csv" -size +10M -delete
