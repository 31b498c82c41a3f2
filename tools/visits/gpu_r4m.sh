#!/bin/bash
# round 4, visit m: the in-process group with one host thread per rank vs one thread for all (M355_GROUP_THREADS=0); its tests
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_shard.py tests/test_streams.py -m gpu -x -q --timeout 600 -k "group or ranks" > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
for t in 1 0 1 0; do
  M355_GROUP_THREADS=$t timeout 600 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end --no-dependent-chain 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); g=d.get('tile_sharded_in_process') or {}
print('M355_GROUP_THREADS=$t: unsharded %.4f ms/pic; group of %s ranks on one GPU: %.4f ms/pic (3 in flight), %.4f non-reference, %.4f one at a time; overhead x%.2f' % (d['ms_per_step'], g.get('ranks'), g.get('ms_per_picture', 0), g.get('ms_per_picture_non_reference', 0), g.get('ms_per_picture_one_at_a_time', 0), g.get('overhead_vs_unsharded') or 0))" | tee -a $OUT/group.txt
done
oracle/_ref/streamgen /tmp/a8k.h265 7680 4320 10 4 2 16 77 5 1 1 >/dev/null 2>&1
for n in 1 2 4 8; do echo "M355_GLUE_RANKS=$n (8K 10-bit 4x2 tiles, dec265 -q -t 8): $(M355_GLUE_RANKS=$n M355_PIPELINE_DEPTH=3 glue/_build/dec265 -q -t 8 /tmp/a8k.h265 2>&1 | grep -o '@ *[0-9.]* fps')"; done | tee $OUT/glue_ranks_one_gpu.txt
