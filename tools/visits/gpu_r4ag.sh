#!/bin/bash
# round 4, visit ag (one box): the round-trip diet — widened picture parameters in the kernel arguments (no vector loads from the argument
# segment), k_sao as ONE round trip, k_deblock's indices in front of the exit, k_residual's loads in one straight line — against the build
# before it (variants/prev.so); hardware parity of the new build on the random / synthetic suites
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4ag; mkdir -p $O
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
run() { # name workload
  v=$1
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth 3 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s %-16s %.4f ms/pic (p10 %.4f p90 %.4f)  one-at-a-time %.4f  %s' % ('$v', '$2', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $O/variants.txt
  unset M355_LIB
}
true
for v in prev base prev base; do run $v c5_8k10_8tiles; done
for v in prev base; do run $v c3_4k_inter; run $v c2_1080p_intra; done
