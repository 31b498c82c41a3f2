#!/bin/bash
# round 5, visit 1: the tree as round 4 left it + every prepared switch on hardware (parity, then same-box A/B).
#   gpurun --timeout 2100 -- 'bash tools/gpu_r5_v1.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v1; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
PAR="tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_gpu_encintra.py tests/test_gpu_pipeline.py tests/test_streams.py"
benchline() { # $1 = label; reads a bench JSON line on stdin
  python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %.4f ms/pic (p10 %.4f p90 %.4f) %s' % ('$1', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"
}
B="python bench.py --no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"

stamp "baseline: whole GPU tier with the opt-in tests on"
M355_TEST_INTRA_ONE_SIDED=1 M355_TEST_FUSE_DBH=1 M355_TEST_DEVICE_WORKLIST=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_all.txt

stamp "parity under each switch"
for sw in M355_INTRA_ONE_SIDED=1 M355_FUSE_DBH=1 M355_DEVICE_WORKLIST=2 M355_DEVICE_WORKLIST=1 M355_MERGE_TU_PLAN=1; do
  env $sw timeout 600 python -m pytest $PAR tests/test_gpu_batch.py tests/test_arena.py -m gpu -x -q 2>&1 | tail -1 | sed "s/^/$sw: /" | tee -a $O/parity.txt
done

stamp "A/B one-sided intra levels"
for rep in 1 2; do for m in 0 1; do for wd in "c2_1080p_intra 1" "c2_1080p_intra 3" "c5_8k10_8tiles 3" "c3_4k_inter 3"; do set -- $wd
  M355_INTRA_ONE_SIDED=$m timeout 200 $B --workload $1 --steps 100 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | benchline "ONE_SIDED=$m $1 depth $2" | tee -a $O/one_sided.txt
done; done; done

stamp "A/B fused H deblock + SAO"
for rep in 1 2; do for m in 0 1; do for wd in "c5_8k10_8tiles 1" "c5_8k10_8tiles 3" "c3_4k_inter 3"; do set -- $wd
  M355_FUSE_DBH=$m timeout 200 $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | benchline "FUSE_DBH=$m $1 depth $2" | tee -a $O/fuse_dbh.txt
done; done; done

stamp "A/B merged TU + plan launch"
for rep in 1 2; do for m in 0 1; do for w in c5_8k10_8tiles c3_4k_inter; do
  M355_MERGE_TU_PLAN=$m timeout 200 $B --workload $w --steps 200 --warmup 10 --pipeline-depth 3 2>>$O/bench.err | benchline "MERGE_TU_PLAN=$m $w depth 3" | tee -a $O/merge.txt
done; done; done

stamp "A/B device work list (submit path)"
for rep in 1 2; do for m in 0 1; do
  M355_DEVICE_WORKLIST=$m timeout 300 python bench.py --no-cpu-baseline --no-dependent-chain --no-end-to-end --steps 20 --warmup 5 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); u=d['with_upload']
print('DEVICE_WORKLIST=$m: with_upload %.4f ms  submit_only %.4f ms  copying %.4f ms  (resident lists %.4f ms)' % (u['ms_per_step'], u['submit_only']['ms_per_step'], u['copying_submit']['ms_per_step'], d['ms_per_step']))" | tee -a $O/worklist.txt
done; done

stamp "driver's line on this box"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2>>$O/bench.err
stamp done
