#!/bin/bash
# round 4, visit j: all GPU tests (incl. the in-process group and M355_GLUE_RANKS), end-to-end attribution, the glue over 4 / 8 ranks on one GPU
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 900 bash tools/e2e_attrib.sh $OUT/e2e_attribution.txt > /dev/null 2>&1; grep -v "m355 glue" $OUT/e2e_attribution.txt
for n in 1 2 4 8; do echo "M355_GLUE_RANKS=$n (8K 10-bit 4x2 tiles, dec265 -q -t 8): $(M355_GLUE_RANKS=$n M355_PIPELINE_DEPTH=3 glue/_build/dec265 -q -t 8 /tmp/a8k.h265 2>&1 | grep -o '@ *[0-9.]* fps')"; done | tee $OUT/glue_ranks_one_gpu.txt
