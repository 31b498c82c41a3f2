#!/bin/bash
# first hardware visit of M355_FUSE_DBH (≈12 GPU-minutes; written in round 4's last session, emulator-verified only): gpurun --timeout 1500 -- 'bash tools/gpu_r5b.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5b; mkdir -p $O
# the horizontal-edge deblocking pass inside the SAO kernel
# (M355_FUSE_DBH=1, k_sao_dbh): parity, then C5 / C3 one and three pictures in flight, alternating; then the kernel trace + PMC traffic of both
M355_TEST_FUSE_DBH=1 timeout 600 python -m pytest tests/test_fuse_dbh.py -m gpu -x -q 2>&1 | tail -3 | sed "s/^/test_fuse_dbh: /" | tee -a $O/parity.txt
M355_FUSE_DBH=1 timeout 900 python -m pytest tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_gpu_encintra.py tests/test_gpu_pipeline.py tests/test_streams.py -m gpu -x -q 2>&1 | tail -1 | sed "s/^/M355_FUSE_DBH=1: /" | tee -a $O/parity.txt
for rep in 1 2; do for m in 0 1; do for w in c5_8k10_8tiles c3_4k_inter; do for depth in 1 3; do
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
