#!/bin/bash
# Where does a picture's wall-clock time go, end to end?  The 8K 10-bit 4x2-tile stream (and the 4K random-access WPP stream) through
# dec265 -q: the reference library, and the glue library in four configurations that add its parts one by one.
# usage: tools/e2e_attrib.sh <outfile>     (M355_GLUE_ATTRIB is a timing switch: pictures are not decoded in those runs)
OUT=${1:-gpurun_out/e2e_attribution.txt}; mkdir -p $(dirname $OUT)
G=oracle/_ref/streamgen; R=oracle/_ref/dec265; M=glue/_build/dec265
fps() { grep -o "@ *[0-9.]* fps" | grep -o "[0-9.]*"; }
{
echo "host cores: $(nproc)   (fps of dec265 -q, 3 runs each; ms = 1000 / best fps)"
$G /tmp/a8k.h265 7680 4320 10 4 2 16 77 5 1 1 >/dev/null 2>&1
$G /tmp/a4k.h265 3840 2160 8 1 1 17 77 5 1 1 $((2048+4096+8192+16384+32768)) >/dev/null 2>&1
for s in a8k a4k; do
  for t in 0 8; do
    echo "== $s -t $t"
    line() { # label env...
      local label=$1; shift; local best=0
      for k in 1 2 3; do env "$@" M355_PIPELINE_DEPTH=3 M355_GLUE_STATS=2 $EXE -q -t $t /tmp/$s.h265 > /tmp/stats.txt 2>&1; f=$(fps < /tmp/stats.txt); best=$(python -c "print(max($best, ${f:-0}))"); done
      printf "%-58s %7.2f fps  %7.2f ms/picture\n" "$label" $best $(python -c "print(1000.0/max($best,1e-9))")
      grep "m355 glue" /tmp/stats.txt | sed 's/^/      /'
    }
    EXE=$R line "reference library (SSE/AVX where it has them)" X=1
    EXE=$M line "glue: parser + recorders only (lists dropped)" M355_GLUE_ATTRIB=1
    EXE=$M line "glue: + list building on the worker (no submit)" M355_GLUE_ATTRIB=2
    EXE=$M line "glue: + submit and decode (the product)" X=1
    EXE=$M line "glue: the same, submit on the decoder's thread" M355_GLUE_SYNC=1
    EXE=$M line "glue: product, glue pool of 4 threads" M355_GLUE_THREADS=4
  done
done
for s in a8k a4k; do
  echo "== $s -t 8 with output (-o /dev/null)"
  for k in 1 2; do echo "reference: $($R -q -t 8 -o /dev/null /tmp/$s.h265 2>&1 | fps) fps    glue: $(M355_PIPELINE_DEPTH=3 $M -q -t 8 -o /dev/null /tmp/$s.h265 2>&1 | fps) fps"; done
done
} > $OUT 2>&1
cat $OUT
