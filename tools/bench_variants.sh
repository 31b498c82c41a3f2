#!/bin/bash
# usage: tools/bench_variants.sh v1 v2 ...   (names under libde265_amd/variants/, "base" = the product build)
for v in "$@"; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$PWD/libde265_amd/variants/$v.so; fi
  python bench.py --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --steps 300 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %.4f ms/pic  %s' % ('$v', d['ms_per_step'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"
done
