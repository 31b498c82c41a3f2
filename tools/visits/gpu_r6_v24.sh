#!/bin/bash
# round 6, visit 24: k_inter_jobs capped at 128 registers (4 waves per SIMD; a few spilled registers in the EDGE path) with and without an LDS pad that keeps it at 3 workgroups per CU
# (then 128 of a SIMD's 512 registers stay free for the other pictures' kernels)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v24; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f verified %s stages %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], d.get('verified'), ' '.join('%s=%.4f' % kv for kv in d['stage_ms'].items())))"; }
run() { if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 300 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/ab.txt; unset M355_LIB; }
for v in base w4 w4pad base w4 w4pad; do run $v c5_8k10_8tiles 3; done
for v in base w4 w4pad base w4 w4pad; do run $v c3_4k_inter 3; done
for v in base w4 w4pad; do run $v c4_4k_4tiles 3; done
