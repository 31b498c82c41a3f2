#!/bin/bash
# round 4, visit aa: full GPU suite after the stage kernels became body + one-picture / batch kernels; driver-like bench; world-1 sharded leg
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/r4aa; mkdir -p $O
( time timeout 1700 python -m pytest tests -m gpu -x -q --timeout 900 > $O/pytest.log 2>&1 ) 2>&1 | grep real; tail -3 $O/pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('driver-like:', d['value'], d['ms_per_step'], d['ms_per_step_spread'], d['stage_ms'], 'frac', d['roofline']['frac'], 'submit_only', d['with_upload']['submit_only'], 'with_upload', d['with_upload']['ms_per_step'], 'e2e', d['end_to_end'].get('speedup'), d['end_to_end']['with_output'].get('speedup'))"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --force-tile-shard --no-cpu-baseline --no-with-upload --no-end-to-end --no-dependent-chain > $O/bench_world1.json 2> $O/bench_world1.err
python -c "
import json; d=json.loads(open('$O/bench_world1.json').read().strip().splitlines()[-1]); print('world 1: unsharded', d['ms_per_step'], 'sharded', json.dumps(d.get('tile_sharded'))[:900])"
