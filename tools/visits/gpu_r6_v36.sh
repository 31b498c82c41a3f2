#!/bin/bash
# round 6, visit 36: blocking frame transfers through pinned staging + a stream of the library (this tree) against the blocking hipMemcpy2D of the tree before
# (libde265_amd/variants/before_stage.so), tools/soak_recheck.py with 32 processes sharing the GPU, pictures 8x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v36; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
echo "== before (blocking hipMemcpy2D, pageable memory, null stream)" | tee $O/soak_recheck_ab.txt
M355_LIB=$GRAFT_REPO_ROOT/libde265_amd/variants/before_stage.so SOAK_SCALE=8 timeout 900 python tools/soak_recheck.py 200000 1600 32 2>&1 | tail -60 | tee -a $O/soak_recheck_ab.txt | cut -c1-200 | head -8
echo "== this tree (pinned staging, copies queued on the library's streams)" | tee -a $O/soak_recheck_ab.txt
SOAK_SCALE=8 timeout 900 python tools/soak_recheck.py 200000 1600 32 2>&1 | tail -60 | tee -a $O/soak_recheck_ab.txt | cut -c1-200 | head -8
SOAK_SCALE=8 timeout 900 python tools/soak_gpu.py 200000 3200 32 2>&1 | tail -30 | tee $O/soak_gpu_scale8.txt | cut -c1-300
SOAK_SCALE=12 timeout 900 python tools/soak_gpu.py 210000 1000 16 2>&1 | tail -30 | tee $O/soak_gpu_scale12.txt | cut -c1-300
