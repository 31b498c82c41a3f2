#!/bin/bash
# picture output beside the decodes: parity + timing with / without the prefetch
OUT=gpurun_out/r03t; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 -k "pipeline or glue or stream or sei or status" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
G=oracle/_ref/streamgen; R=oracle/_ref/dec265; M=glue/_build/dec265
$G /tmp/s8k10.h265 7680 4320 10 4 2 16 77 5 1 1 > /dev/null 2>&1
$G /tmp/s4k10.h265 3840 2160 10 2 2 24 77 5 1 1 > /dev/null 2>&1
$G /tmp/s1080p.h265 1920 1080 8 2 2 40 77 5 1 1 > /dev/null 2>&1
for s in s8k10 s4k10 s1080p; do
  for rep in 1 2; do
    echo "== $s reference -t 8 -o /dev/null: $($R -q -t 8 -o /dev/null /tmp/$s.h265 2>&1 | tail -1)"
    echo "== $s glue prefetch   -o /dev/null: $(M355_PIPELINE_DEPTH=3 $M -q -t 8 -o /dev/null /tmp/$s.h265 2>&1 | tail -1)"
    echo "== $s glue noprefetch -o /dev/null: $(M355_GLUE_NO_PREFETCH=1 M355_PIPELINE_DEPTH=3 $M -q -t 8 -o /dev/null /tmp/$s.h265 2>&1 | tail -1)"
    echo "== $s glue no output            : $(M355_PIPELINE_DEPTH=3 $M -q -t 8 /tmp/$s.h265 2>&1 | tail -1)"
  done
  $R -q -t 8 -o /tmp/r.yuv /tmp/$s.h265 > /dev/null 2>&1; M355_PIPELINE_DEPTH=3 M355_GLUE_STATS=1 $M -q -t 8 -o /tmp/m.yuv /tmp/$s.h265 2>&1 | grep "m355 glue"
  echo "md5 $s ref $(md5sum < /tmp/r.yuv | cut -c1-32) glue $(md5sum < /tmp/m.yuv | cut -c1-32)"
done 2>&1 | tee $OUT/output_prefetch.txt
