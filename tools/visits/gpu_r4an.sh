#!/bin/bash
# round 4, visit an (final state of the round): full GPU suite, the driver's bench command, rocprofv3 kernel trace of the same workload
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4an; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 > $O/pytest.log 2>&1 ) 2>&1 | grep real; grep -E "passed|failed|error" $O/pytest.log | tail -2
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('driver-like:', d['value'], d['ms_per_step'], d['ms_per_step_spread'], d['stage_ms'], 'frac', d['roofline']['frac'], 'submit_only', d['with_upload']['submit_only'], 'with_upload', d['with_upload']['ms_per_step'], 'e2e', d['end_to_end'].get('speedup'), d['end_to_end']['with_output'].get('speedup'), 'cpu', d['cpu_baseline']['value'])"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $REPO/bench.py --workload c5_8k10_8tiles --steps 60 --warmup 5 --pipeline-depth 1 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end > $O/kt_bench.json 2> $O/kt.log
cd $REPO; python tools/rocprof_summary.py $O/kt $O/kernel_stats_c5.txt | head -16 | cut -c1-150
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete
