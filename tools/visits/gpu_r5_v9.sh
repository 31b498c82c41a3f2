#!/bin/bash
# round 5, visit 9: with the lean k_inter_jobs, does the FUSED residual order (tiles added in the write-back) now win with pictures in flight?  128-lane workgroups?
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v9; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain --no-cold-refs"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-14s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2 3; do for v in "base 0" "base 1" "blk128 0"; do set -- $v; vv=$1; fz=$2
 for wd in "c5_8k10_8tiles 3" "c3_4k_inter 3" "c4_4k_4tiles 3"; do set -- $wd
  if [ "$vv" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$vv.so; fi
  M355_RES_FUSED=$fz timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line "$vv/fused=$fz" $1 $2 | tee -a $O/fused_blk.txt
done; done; done
