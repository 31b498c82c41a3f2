#!/bin/bash
# round 4, visit aq (last tree): hardware parity after the k_intra_plan grid change, pictures in flight re-swept with the lighter packets
# (C5 / C3: depth 2..5), the driver's command once more
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4aq; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_gpu_encintra.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -1 | tee $O/parity.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"
for w in c5_8k10_8tiles c3_4k_inter; do for d in 2 3 4 5; do
  timeout 200 python bench.py $B --workload $w --steps 200 --warmup 10 --pipeline-depth $d 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-16s depth $d: %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f  %s' % ('$w', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $O/depth.txt
done; done
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('driver-like:', d['value'], d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'], 'submit_only', d['with_upload']['submit_only']['ms_per_step'], 'with_upload', d['with_upload']['ms_per_step'], 'e2e', d['end_to_end'].get('speedup'), d['end_to_end']['with_output'].get('speedup'))"
