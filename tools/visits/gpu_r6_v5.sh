#!/bin/bash
# round 6, visit 5: the per-lane k_inter tree (VOP3P heads) — whole GPU tier; the bench line with the new columns (copy ceiling, dec265 -0 / -t 0, full-size CPU baseline, frame checks);
# what an idle RCCL communicator costs and why (tools/rccl_idle_ab.py)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v5; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "GPU tier"
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
stamp "driver's command"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench.err ) 2>&1 | grep real | tee -a $O/timeline.txt
python -c "
import json; d=json.loads(open('$O/bench_driver_line.json').read().strip().splitlines()[-1]); print('driver-like:', d['value'], d['ms_per_step'], d['stage_ms']); print('roofline', json.dumps(d['roofline'])[:500]); print('verified', d.get('verified'), 'e2e', json.dumps(d.get('end_to_end'))[:900]); print('cpu', json.dumps(d.get('cpu_baseline'))[:1200])" | tee -a $O/timeline.txt
tail -5 $O/bench.err
stamp "idle RCCL communicator"
timeout 900 python tools/rccl_idle_ab.py c5_8k10_8tiles 100 2>&1 | tee $O/rccl_idle_ab.txt
stamp done
