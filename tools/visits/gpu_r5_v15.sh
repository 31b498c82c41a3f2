#!/bin/bash
# round 5, visit 15: k_intra on intra pictures — levels from the picture-wide clock (runtime_upload.hip intra_schedule) + halo keeper at depth 1,
# against: asap (depth-only levels), nokeep (no keeper wave), kt150 (keeper asks again every 1.5 us, not every level)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v15; mkdir -p $O
timeout 600 python -m pytest tests/test_intra_halo_late.py tests/test_intra_one_sided.py tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_encintra_streams.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/parity.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2; do for v in asap nokeep kt150 base; do for wd in "c2_1080p_intra 1" "c2_1080p_intra 3"; do set -- $wd
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload $1 --steps 100 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line $v $1 $2 | tee -a $O/intra_timed_ab.txt
done; done; done
unset M355_LIB
# the batch path (32 in flight, one intra stage per 8 pictures) and the host side of an intra picture's submit
timeout 300 python bench.py $B --workload c2_1080p_intra --steps 96 --warmup 8 --pipeline-depth 32 --intra-batch 8 2>>$O/bench.err | line base c2_batch 32x8 | tee -a $O/intra_timed_ab.txt
for v in asap base; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  M355_PROFILE_UPLOAD=1 timeout 120 python - 2>&1 <<'P' | grep "m355 upload" | tail -2 | sed "s/^/$v: /" | tee -a $O/intra_timed_ab.txt
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from libde265_amd import capi, synth, worklist
lib = capi.Library(); ctx = capi.Context(lib, 0)
pic = synth.picture(**synth.CONFIGS['c2_1080p_intra']); pp = pic.pp[0]
pic.dst_frame = ctx.frame_create_for(pp); pic.ref_frames = [-1] * worklist.MAX_REF_FRAMES
for i in range(6):
    h = ctx.upload(pic); ctx.wait(); ctx.release(h)
P
done
unset M355_LIB
for wd in "c3_4k_inter 3" "c5_8k10_8tiles 3"; do set -- $wd
  timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line base $1 $2 | tee -a $O/intra_timed_ab.txt
done
