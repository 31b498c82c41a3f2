#!/bin/bash
# round 4, visit z: how much does residency cost the sparse k_intra (inter pictures)?  LDS padded by 4 / 8 / 16 / 32 KB per workgroup
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/r4z; mkdir -p $O
for v in base lds_p4 lds_p8 lds_p16 lds_p32 base; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  for w in c5_8k10_8tiles c3_4k_inter; do
  timeout 300 python bench.py --no-cpu-baseline --no-with-upload --no-end-to-end --no-dependent-chain --workload $w --steps 200 --warmup 10 --pipeline-depth 3 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-15s depth 3: %.4f ms/pic  one-at-a-time %.4f  intra=%.4f' % ('$v', '$w', d['ms_per_step'], d['ms_per_step_one_in_flight'], d['stage_ms']['intra']))" | tee -a $O/summary.txt
  done
done
