#!/bin/bash
# round 5, visit 32: the last tree — whole GPU tier, the driver's command, C3 / C4 with their chains
#   gpurun --timeout 600 -- 'bash tools/visits/gpu_r5_v32.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v32; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_driver_line.json').read().strip().splitlines()[-1]); print('driver-like:', d['value'], d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'], 'chain', d['dependent_chain']['ms_per_step'])" | tee $O/summary.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs"
for w in c3_4k_inter c4_4k_4tiles; do timeout 100 python bench.py $B --workload $w --steps 200 --warmup 10 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], 'one at a time', d['ms_per_step_one_in_flight'], 'chain', d['dependent_chain']['ms_per_step'], d['stage_ms'])" | tee -a $O/summary.txt; done
