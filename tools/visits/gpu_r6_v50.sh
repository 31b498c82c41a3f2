#!/bin/bash
# round 6, visit 50: the last tree once more — smoke(), the GPU tier, the driver's bench command
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v50; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_driver_line.json').read().strip().splitlines()[-1]); print('driver line:', d['value'], d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'], 'verified', d['verified'], d['dependent_chain']['verified'], d['rotating_references']['verified'], d['with_upload']['verified'])" | tee $O/line.txt
