#!/bin/bash
# round 4, visit ak: ONE mark per decode (EvRef ring) instead of an event per object — full GPU suite on it, then A/B against the build
# before it (variants/prev2.so = commit 69591ea), C5 / C3 / C4 three in flight, the dependent chain and the submit path
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4ak; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest.log 2>&1 ) 2>&1 | grep real; tail -3 $O/pytest.log
B="--no-cpu-baseline --no-end-to-end"
for rep in 1 2; do for v in prev2 base; do for w in c5_8k10_8tiles c3_4k_inter c4_4k_4tiles; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload $w --steps 200 --warmup 10 --pipeline-depth 3 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); u=d.get('with_upload') or {}; ch=d.get('dependent_chain') or {}
print('%-6s %-16s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f host enqueue %.4f  chain %.4f  submit_only %.4f with_upload %.4f' % ('$v', '$w', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], d['host_enqueue_ms_per_step'], ch.get('ms_per_step',0), u.get('submit_only',{}).get('ms_per_step',0), u.get('ms_per_step',0)))" | tee -a $O/marks.txt
  unset M355_LIB
done; done; done
