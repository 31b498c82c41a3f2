#!/bin/bash
OUT=gpurun_out/${1:-r02c}; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 300 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for w in c2_1080p_intra c3_4k_inter c4_4k_4tiles c5_8k10_8tiles; do
  for d in 1 3; do timeout 300 python bench.py --workload $w --no-cpu-baseline --pipeline-depth $d > $OUT/bench_${w}_d$d.json 2>$OUT/err.txt; python - <<PY
import json
try:
  d=json.load(open("$OUT/bench_${w}_d$d.json"))
  print("$w depth $d", round(d["ms_per_step"],4), round(d.get("ms_per_step_one_in_flight",0),4), d["stage_ms"])
except Exception as e: print("$w depth $d FAILED", e)
PY
  done
done
