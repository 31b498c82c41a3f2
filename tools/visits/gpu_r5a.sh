#!/bin/bash
# first hardware visit of M355_DEVICE_WORKLIST (k_intra's work list made on the device: k_work_keys / k_work_items, written at the end of
# round 4 when the round's GPU minutes were spent): parity (mode 2 = library-internal comparison with the host's list for every upload,
# mode 1 = decode from the device-made list), then the submit path with and without it (the host phases it takes off the submitting thread)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5a; mkdir -p $O
M355_TEST_DEVICE_WORKLIST=1 timeout 600 python -m pytest tests/test_device_worklist.py -m gpu -x -q 2>&1 | tail -3 | tee $O/parity.txt
for m in 2 1; do M355_DEVICE_WORKLIST=$m timeout 600 python -m pytest tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_arena.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -1 | sed "s/^/M355_DEVICE_WORKLIST=$m: /" | tee -a $O/parity.txt; done
for rep in 1 2 3; do for m in 0 1; do
  M355_DEVICE_WORKLIST=$m timeout 300 python bench.py --no-cpu-baseline --no-dependent-chain --no-end-to-end --steps 20 --warmup 5 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); u=d['with_upload']
print('M355_DEVICE_WORKLIST=$m: with_upload %.4f ms  submit_only %.4f ms  copying %.4f ms  (resident lists %.4f ms)' % (u['ms_per_step'], u['submit_only']['ms_per_step'], u['copying_submit']['ms_per_step'], d['ms_per_step']))" | tee -a $O/submit.txt
done; done
M355_DEVICE_WORKLIST=1 M355_PROFILE_UPLOAD=1 timeout 120 python tools/prof_submit.py 2>&1 | tail -6 | tee $O/submit_host_phases.txt
# second prepared experiment: transform edges + border plans in one launch (M355_MERGE_TU_PLAN=1): parity, then C5 / C3 three in flight, alternating
M355_MERGE_TU_PLAN=1 timeout 600 python -m pytest tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_gpu_encintra.py -m gpu -x -q 2>&1 | tail -1 | sed "s/^/M355_MERGE_TU_PLAN=1: /" | tee -a $O/parity.txt
for rep in 1 2 3; do for m in 0 1; do for w in c5_8k10_8tiles c3_4k_inter; do
  M355_MERGE_TU_PLAN=$m timeout 200 python bench.py --no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain --workload $w --steps 200 --warmup 10 --pipeline-depth 3 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('M355_MERGE_TU_PLAN=$m %-16s %.4f ms/pic (p10 %.4f p90 %.4f)' % ('$w', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90']))" | tee -a $O/merge.txt
done; done; done
# fourth / fifth: tools/gpu_r5b.sh (M355_FUSE_DBH: the horizontal deblocking pass inside the SAO kernel) and tools/gpu_r5c.sh (M355_INTRA_ONE_SIDED:
# mode-aware dependency levels for k_intra) — visits of their own, ≈12 GPU-minutes each
# third: the prepared patches — apply each to a copy of the tree, tools/variants.sh <name> "", then tools/bench_variants.sh base <name> (C5) and the same for c3_4k_inter:
#   tools/experiments/inter_prologue_overlap.patch, sao_saddr_offsets.patch, meta_sao_batched_neighbour_loads.patch
