#!/bin/bash
# round 5, visit 11: (1) k_inter_jobs capped to 2 workgroups per CU / 128-lane workgroups; (2) the dependent chain with and without the fused residual order
# (the library of the commit before its removal as a variant: does taking k_residual off a waiting picture's critical path pay when every picture waits for its references?)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v11; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
c=d.get('dependent_chain') or {}
print('%-14s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f chain %.4f %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], c.get('ms_per_step', 0), ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2 3; do for v in "base x" "pad2 x" "blk128 x" "withfused 0" "withfused 1"; do set -- $v; vv=$1; fz=$2
 for wd in "c5_8k10_8tiles 3" "c3_4k_inter 3"; do set -- $wd
  if [ "$vv" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$vv.so; fi
  if [ "$fz" = x ]; then unset M355_RES_FUSED; else export M355_RES_FUSED=$fz; fi
  timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line "$vv/$fz" $1 $2 | tee -a $O/chain_fused_pad.txt
done; done; done
unset M355_LIB M355_RES_FUSED
cd /tmp
for v in base pad2; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --pmc $c -d $O/${v}_$c -o x --output-format csv -- python $REPO/bench.py --workload c5_8k10_8tiles --steps 5 --warmup 1 $B --no-dependent-chain --pipeline-depth 1 > $O/${v}_$c.log 2>&1; done
  python $REPO/tools/pmc_summary.py $O/${v}_FETCH_SIZE $O/${v}_WRITE_SIZE 2>&1 | grep -E "kernel|k_inter" | cut -c1-110 | sed "s/^/$v: /" | tee -a $O/chain_fused_pad.txt
done
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +5M -delete
