#!/bin/bash
# round 5, visit 5: k_inter_jobs per-workgroup timeline (C3, C5) + the tables-from-memory build on this box
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v5; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"
for w in c3_4k_inter c5_8k10_8tiles; do M355_LIB=$REPO/libde265_amd/variants/prof.so timeout 120 python tools/prof_inter_timeline.py $w 2>&1 | tail -30 | tee -a $O/inter_timeline.txt; done
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2; do for wd in "c5_8k10_8tiles 3" "c3_4k_inter 3" "c4_4k_4tiles 3"; do set -- $wd
  timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line base $1 $2 | tee -a $O/tables_from_memory.txt
done; done
timeout 300 python -m pytest tests/test_inter_extremes.py tests/test_gpu_random.py tests/test_gpu_synth.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/parity.txt
