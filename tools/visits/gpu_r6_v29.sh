#!/bin/bash
# round 6, visit 29: the 11 streams of visit 28's soak that came out different on the hardware (all 4:2:2, tile columns, inter pictures): where, how often, with one lane
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v29; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
SEEDS="240 594 259 307 404 468 229 70 166 28 588"
echo "== default" | tee $O/diag.txt
timeout 600 python tools/diag_stream.py 3 $SEEDS 2>&1 | tee -a $O/diag.txt
echo "== M355_PIPELINE_DEPTH=1" | tee -a $O/diag.txt
M355_PIPELINE_DEPTH=1 timeout 600 python tools/diag_stream.py 2 594 307 70 2>&1 | tee -a $O/diag.txt
echo "== M355_GLUE_SYNC=1" | tee -a $O/diag.txt
M355_GLUE_SYNC=1 timeout 600 python tools/diag_stream.py 2 594 307 70 2>&1 | tee -a $O/diag.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5 | tee $O/pytest_all.txt
