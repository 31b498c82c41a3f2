#!/bin/bash
# round 6, visit 52: XCD-contiguous block orders of k_sao (runs of two block rows) and k_residual (each size bin as eight contiguous runs) again, on this round's pipeline
# (round 4 measured them at 0.385 ms per picture: a third less fabric fetch for both kernels, no time gained) — variants saoxcd / resxcd / bothxcd against the product
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v52; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
M355_LIB=$GRAFT_REPO_ROOT/libde265_amd/variants/bothxcd.so timeout 600 python -m pytest tests/test_gpu_random.py tests/test_gpu_synth.py -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1 | tee $O/parity_bothxcd.txt
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --no-cold-refs"
for rep in 1 2 3; do for v in product saoxcd resxcd bothxcd; do
  L=; [ $v != product ] && L=$GRAFT_REPO_ROOT/libde265_amd/variants/$v.so
  M355_LIB=$L timeout 200 python bench.py $B --workload c5_8k10_8tiles --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v: %.4f ms/picture (p10 %.4f p90 %.4f), one at a time %.4f, stages %s verified %s' % (d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], d['stage_ms'], d['verified']))" | tee -a $O/xcd_ab.txt
done; done
