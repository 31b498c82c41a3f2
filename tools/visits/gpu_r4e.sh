#!/bin/bash
# round 4, visit e: the RCCL transport on hardware (one rank loopback + two rank processes on the one GPU), host phases of the submit
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rccl.py -x -q -rs --timeout 600 > $OUT/pytest_rccl.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_rccl.log; tail -8 $OUT/pytest_rccl.log
cp gpurun_out/rccl_two_ranks_one_gpu.json $OUT/ 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/rccl_two_ranks_one_gpu.json')); print(json.dumps(d['results'])[:1500]); print(d['logs'][0][-1500:])" 2>/dev/null
M355_PROFILE_UPLOAD=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dependent-chain --no-end-to-end > $OUT/bench_upload.json 2> $OUT/upload_profile.txt
grep "in place" $OUT/upload_profile.txt | tail -30 | awk '{print}' > $OUT/upload_phases.txt; tail -5 $OUT/upload_phases.txt
grep "m355 submit" $OUT/upload_profile.txt | tail -5
python -c "
import json; d=json.loads(open('$OUT/bench_upload.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['with_upload'])"
