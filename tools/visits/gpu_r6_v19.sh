#!/bin/bash
# round 6, visit 19: k_sao's output stores plain instead of non-temporal — does the frame a chain's next picture references then wait in the Infinity Cache?
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v19; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-verify"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
rot=d.get('rotating_references') or {}; ch=d.get('dependent_chain') or {}
print('%-9s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f rotating %.4f chain %.4f stages %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], rot.get('ms_per_step') or 0, ch.get('ms_per_step') or 0, ' '.join('%s=%.4f' % kv for kv in d['stage_ms'].items())))"; }
run() { # variant workload depth
  if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 300 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/sao_plain_ab.txt
  unset M355_LIB
}
for v in base sao_plain base sao_plain; do run $v c5_8k10_8tiles 3; done
for v in base sao_plain base sao_plain; do run $v c4_4k_4tiles 3; done
