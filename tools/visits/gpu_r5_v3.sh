#!/bin/bash
# round 5, visit 3: where does k_inter_jobs' time go?  Attribution builds of the lean kernel (no loads / cache-hot loads / no stores /
# no PB plane / workgroup sizes) on ONE box, one picture at a time and three in flight; the MFMA micro-benchmark with its constants from a table.
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v3; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"
timeout 300 tools/ubench/_build/ub_mfma_idct > $O/ub_mfma_idct.txt 2>&1; cat $O/ub_mfma_idct.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2; do for v in base noload hot nostore hot_nostore compute nopbof blk128 blk64; do for wd in "c5_8k10_8tiles 3" "c3_4k_inter 3"; do set -- $wd
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line $v $1 $2 | tee -a $O/inter_attribution.txt
done; done; done
unset M355_LIB
