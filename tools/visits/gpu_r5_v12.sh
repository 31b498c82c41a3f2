#!/bin/bash
# round 5, visit 12: deblocking as ONE pass over shifted tiles (k_deblock_tiles) against the two-pass kernels (-DM355_DEBLOCK_TWO_PASS): parity, then A/B on one box
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v12; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d.get('dependent_chain') or {}
print('%-8s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f chain %.4f %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], c.get('ms_per_step', 0), ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2 3; do for v in twopass base; do for wd in "c5_8k10_8tiles 3" "c3_4k_inter 3" "c4_4k_4tiles 3"; do set -- $wd
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line $v $1 $2 | tee -a $O/deblock_tiles_ab.txt
done; done; done
unset M355_LIB
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --pmc $c -d $O/t_$c -o x --output-format csv -- python $REPO/bench.py --workload c5_8k10_8tiles --steps 5 --warmup 1 $B --no-dependent-chain --pipeline-depth 1 > $O/t_$c.log 2>&1; done
python $REPO/tools/pmc_summary.py $O/t_FETCH_SIZE $O/t_WRITE_SIZE 2>&1 | grep -E "kernel|k_deblock" | cut -c1-110 | tee -a $O/deblock_tiles_ab.txt
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +5M -delete
