#!/bin/bash
# round 6, visit 42: the GPU tier UNDER LOAD — the slot-layer tests on 24 processes at once (x3), then the whole tier on 8 processes sharing the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v42; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
for i in 1 2 3; do
  for k in $(seq 1 24); do (timeout 600 python -m pytest tests/test_gpu_slots.py tests/test_slot_layer_live.py -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1 > $O/slots_${i}_${k}.txt) & done
  wait
done
cat $O/slots_*.txt | sort | uniq -c | tee $O/slots_under_load.txt
rm -f $O/slots_?_*.txt
timeout 1500 python -m pytest tests -q -m gpu -n 8 -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -15 | tee $O/pytest_n8.txt
