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
# fourth (written in the round's last session, emulator-verified only): the horizontal-edge deblocking pass inside the SAO kernel
# (M355_FUSE_DBH=1, k_sao_dbh): parity, then C5 / C3 one and three pictures in flight, alternating; then the kernel trace + PMC traffic of both
M355_TEST_FUSE_DBH=1 timeout 600 python -m pytest tests/test_fuse_dbh.py -m gpu -x -q 2>&1 | tail -3 | sed "s/^/test_fuse_dbh: /" | tee -a $O/parity.txt
M355_FUSE_DBH=1 timeout 900 python -m pytest tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_gpu_encintra.py tests/test_gpu_pipeline.py tests/test_streams.py -m gpu -x -q 2>&1 | tail -1 | sed "s/^/M355_FUSE_DBH=1: /" | tee -a $O/parity.txt
for rep in 1 2 3; do for m in 0 1; do for w in c5_8k10_8tiles c3_4k_inter; do for depth in 1 3; do
  M355_FUSE_DBH=$m timeout 200 python bench.py --no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain --workload $w --steps 200 --warmup 10 --pipeline-depth $depth 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('M355_FUSE_DBH=$m %-16s depth $depth %.4f ms/pic (p10 %.4f p90 %.4f) %s' % ('$w', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $O/fuse_dbh.txt
done; done; done; done
cd /tmp; for m in 0 1; do
  M355_FUSE_DBH=$m timeout 300 rocprofv3 --kernel-trace --stats -d $O/dbh_trace_$m -- python $REPO/bench.py --no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain --steps 50 --warmup 5 --pipeline-depth 1 > /dev/null 2>>$O/bench.err
  f=$(find $O/dbh_trace_$m -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep -E "k_sao|k_deblock|Name" "$f" | cut -c1-200 | sed "s/^/M355_FUSE_DBH=$m: /" | tee -a $O/fuse_dbh.txt
  M355_FUSE_DBH=$m timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE -d $O/dbh_pmc_$m -- python $REPO/bench.py --no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain --steps 10 --warmup 2 --pipeline-depth 1 > /dev/null 2>>$O/bench.err
done; cd $REPO
# fifth (same session, emulator-verified only): k_intra's dependency levels from what each intra mode can read (M355_INTRA_ONE_SIDED=1:
# 42 % fewer levels on the C2 picture): parity, then C2 one picture at a time / three in flight and C5, alternating
M355_TEST_INTRA_ONE_SIDED=1 timeout 600 python -m pytest tests/test_intra_one_sided.py -m gpu -x -q 2>&1 | tail -3 | sed "s/^/test_intra_one_sided: /" | tee -a $O/parity.txt
M355_INTRA_ONE_SIDED=1 timeout 900 python -m pytest tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_gpu_encintra.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py tests/test_streams.py -m gpu -x -q 2>&1 | tail -1 | sed "s/^/M355_INTRA_ONE_SIDED=1: /" | tee -a $O/parity.txt
for rep in 1 2 3; do for m in 0 1; do for wd in "c2_1080p_intra 1" "c2_1080p_intra 3" "c5_8k10_8tiles 3" "c3_4k_inter 3"; do set -- $wd
  M355_INTRA_ONE_SIDED=$m timeout 200 python bench.py --no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain --workload $1 --steps 100 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('M355_INTRA_ONE_SIDED=$m %-16s depth $2 %.4f ms/pic (p10 %.4f p90 %.4f) %s' % ('$1', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $O/one_sided.txt
done; done; done
# third: the prepared patches — apply each to a copy of the tree, tools/variants.sh <name> "", then tools/bench_variants.sh base <name> (C5) and the same for c3_4k_inter:
#   tools/experiments/inter_prologue_overlap.patch, sao_saddr_offsets.patch, meta_sao_batched_neighbour_loads.patch
