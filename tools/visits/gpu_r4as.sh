#!/bin/bash
# round 4, visit as (the round's last tree): full GPU suite + the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4as; mkdir -p $O
( time timeout 140 python -m pytest tests -m gpu -x -q --timeout 120 > $O/pytest.log 2>&1 ) 2>&1 | grep real; grep -E "passed|failed|error" $O/pytest.log | tail -3
( time timeout 70 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('driver-like:', d['value'], d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'], 'submit_only', d['with_upload']['submit_only']['ms_per_step'], 'with_upload', d['with_upload']['ms_per_step'], 'e2e', d['end_to_end'].get('speedup'), d['end_to_end']['with_output'].get('speedup'))"
