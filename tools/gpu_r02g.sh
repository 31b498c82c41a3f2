#!/bin/bash
OUT=gpurun_out/${1:-r02g}; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 300 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
( time M355_PROFILE_UPLOAD=1 timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -3
tail -c 4000 $OUT/bench.json; grep "m355 upload" $OUT/bench.err | tail -3
