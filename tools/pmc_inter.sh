#!/bin/bash
# k_inter_jobs of one or more kernel variants: event time (bench stage_ms) + fabric traffic (separate PMC passes).
# usage: tools/pmc_inter.sh base|<variant> ...     -> gpurun_out/pmc_inter/<variant>.txt
REPO=$PWD; export TMPDIR=/tmp
FL="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
for v in "$@"; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  OUT=$REPO/gpurun_out/pmc_inter/$v; rm -rf $OUT; mkdir -p $OUT
  cd $REPO; python bench.py $FL --steps 300 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v %.4f ms/pic  %s' % (d['ms_per_step'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee $OUT.txt
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    n=$(echo $c | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $c -d $OUT/$n -o x --output-format csv -- python $REPO/bench.py $FL --steps 4 --warmup 1 --pipeline-depth 1 > $OUT/$n.log 2>&1
  done
  python $REPO/tools/pmc_summary.py $OUT 2>&1 | grep -E "^kernel|k_inter|k_meta_pb" | tee -a $OUT.txt
  rm -rf $OUT
done
