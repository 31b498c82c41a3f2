#!/bin/bash
# round 4, visit b: GPU tests (incl. the random-access streams), the driver's bench line, end to end for C3 (RA + WPP stream) and C5
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
tail -c 3000 $OUT/bench.json
timeout 600 python bench.py --workload c3_4k_inter --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_c3.json 2>> $OUT/bench.err
python - <<PY
import json
for f in ("bench.json", "bench_c3.json"):
    try:
        d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d.get("with_upload", {}).get("submit_only"), json.dumps(d.get("end_to_end"))[:900])
    except Exception as e:
        print(f, "unreadable", e)
PY
