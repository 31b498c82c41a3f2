#!/bin/bash
# End-to-end decode of synthetic streams: the reference CLI on the reference library (SSE/AVX where it has them) vs the SAME CLI
# on glue/_build/libde265.so (pixels on the GPU).  usage: tools/e2e.sh <outdir>
OUT=${1:-gpurun_out/e2e}; mkdir -p $OUT
G=oracle/_ref/streamgen; R=oracle/_ref/dec265; M=glue/_build/dec265
echo "host cores: $(nproc)" > $OUT/e2e.txt
run() { # name W H bd tc tr frames
  local n=$1; shift
  ( time $G /tmp/$n.h265 $1 $2 $3 $4 $5 $6 77 5 1 1 ) 2>&1 | grep real | sed "s/^/gen $n: /" >> $OUT/e2e.txt
  ls -la /tmp/$n.h265 | awk '{print "bytes:", $5}' >> $OUT/e2e.txt
  for t in 0 8 32; do
    echo "== $n reference dec265 -t $t" >> $OUT/e2e.txt; $R -q -t $t /tmp/$n.h265 2>&1 | tail -1 >> $OUT/e2e.txt
    echo "== $n MI355X glue dec265 -t $t" >> $OUT/e2e.txt; M355_GLUE_STATS=1 M355_PIPELINE_DEPTH=3 $M -q -t $t /tmp/$n.h265 2>&1 | tail -2 >> $OUT/e2e.txt
  done
  $R -q -t 8 -o /tmp/r.yuv /tmp/$n.h265 > /dev/null 2>&1; $M -q -t 8 -o /tmp/m.yuv /tmp/$n.h265 > /dev/null 2>&1
  echo "md5 ref $(md5sum < /tmp/r.yuv | cut -c1-32) glue $(md5sum < /tmp/m.yuv | cut -c1-32)" >> $OUT/e2e.txt
}
run s1080p 1920 1080 8 2 2 40
run s4k10 3840 2160 10 2 2 24
run s8k10 7680 4320 10 4 2 16
cat $OUT/e2e.txt
