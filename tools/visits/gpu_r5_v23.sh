#!/bin/bash
# round 5, visit 23: the sparse prologue's need scan in registers + residual loads across the (LDS-only) barrier + halo loads that do not wait for the CU
# records (base, 80 registers) against visit 22's tree (v22) and the 87-register build (w5); tufork = transform edges + border plans on the side stream
# of a one-stream lane; prologue stamps of the new tree
#   gpurun --timeout 900 -- 'bash tools/gpu_r5_v23.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v23; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "parity"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
for v in tufork w5; do M355_LIB=$REPO/libde265_amd/variants/$v.so timeout 300 python -m pytest tests/test_gpu_synth.py tests/test_gpu_random.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -1 | sed "s/^/$v: /" | tee -a $O/pytest_all.txt; done
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
dc=d.get('dependent_chain') or {}
print('%-6s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f chain %s %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ('%.4f' % dc['ms_per_step']) if 'ms_per_step' in dc else '-', ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
run() { # variant workload depth
  if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 200 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/intra_prologue_ab.txt
  unset M355_LIB
}
stamp "A/B"
for wd in "c5_8k10_8tiles 3" "c3_4k_inter 3"; do set -- $wd; for v in v22 base w5 tufork tufork w5 base v22; do run $v $1 $2; done; done
for wd in "c5_8k10_8tiles 1" "c3_4k_inter 1" "c4_4k_4tiles 3" "c4_4k_4tiles 1"; do set -- $wd; for v in v22 base tufork; do run $v $1 $2; done; done
stamp "timelines"
for w in c3_4k_inter c5_8k10_8tiles; do for v in prof1 prof2 prof3 prof4 prof; do echo "== $v" | tee -a $O/intra_prologue_stamps.txt; M355_LIB=$REPO/libde265_amd/variants/$v.so timeout 120 python tools/prof_timeline_sparse.py $w 2>&1 | grep -E "CTBs with|prologue|block loop|whole CTB" | tee -a $O/intra_prologue_stamps.txt; done; done
stamp done
