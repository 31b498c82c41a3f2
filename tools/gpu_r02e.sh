#!/bin/bash
OUT=gpurun_out/${1:-r02e}; mkdir -p $OUT
for nw in 3 4 6; do
for w in c3_4k_inter c5_8k10_8tiles; do
  M355_INTRA_NW=$nw timeout 300 python bench.py --workload $w --no-cpu-baseline --pipeline-depth 1 > $OUT/bench_${w}_nw$nw.json 2>$OUT/err.txt; python - <<PY
import json
try:
  d=json.load(open("$OUT/bench_${w}_nw$nw.json"))
  print("NW=$nw $w", round(d["ms_per_step"],4), d["stage_ms"])
except Exception as e: print("$w FAILED", e)
PY
done; done
timeout 300 python bench.py --workload c2_1080p_intra --no-cpu-baseline --pipeline-depth 1 > $OUT/bench_c2.json 2>$OUT/err.txt; python -c "
import json; d=json.load(open('$OUT/bench_c2.json')); print('c2', d['ms_per_step'], d['stage_ms'])"
