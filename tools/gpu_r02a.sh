#!/bin/bash
# round 2, visit a: parity tests (incl. the live decode + slot-layer tests), end-to-end timing of the glue decoder vs the reference CLI
OUT=$PWD/gpurun_out/r02a; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
S=glue/_build/testdata/girlshy.h265
nproc > $OUT/e2e.txt
for t in 0 4 16; do
  echo "== glue dec265 -t $t" >> $OUT/e2e.txt
  ( time M355_GLUE_STATS=1 glue/_build/dec265 -q -t $t -o /tmp/g.yuv $S ) >> $OUT/e2e.txt 2>&1
  md5sum /tmp/g.yuv >> $OUT/e2e.txt
  echo "== reference dec265 -t $t (SSE/AVX)" >> $OUT/e2e.txt
  ( time oracle/_ref/dec265 -q -t $t -o /tmp/r.yuv $S ) >> $OUT/e2e.txt 2>&1
  md5sum /tmp/r.yuv >> $OUT/e2e.txt
done
cat $OUT/e2e.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 2500 $OUT/bench.json
