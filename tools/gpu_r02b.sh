OUT=gpurun_out/r02b; mkdir -p $OUT
for w in c2_1080p_intra c3_4k_inter c4_4k_4tiles; do
  for d in 1 3; do timeout 300 python bench.py --workload $w --no-cpu-baseline --pipeline-depth $d > $OUT/bench_${w}_d$d.json 2>$OUT/err.txt; python - <<PY
import json
d=json.load(open("$OUT/bench_${w}_d$d.json"))
print("$w depth $d", d["ms_per_step"], d.get("ms_per_step_one_in_flight"), d["stage_ms"])
PY
  done
done
