#!/bin/bash
# round 6, visit 30: the glue takes the chroma block's rotation look-up when the whole picture is parsed — the 11 streams of visit 28 again (each five times with threads),
# then the stream soak over 3000 further seeds and the GPU tier's stream tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v30; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
timeout 600 python tools/diag_stream.py 5 240 594 259 307 404 468 229 70 166 28 588 2>&1 | grep -c "identical" | tee $O/diag_identical_count.txt
timeout 600 python tools/diag_stream.py 5 240 594 259 307 404 468 229 70 166 28 588 2>&1 | grep -v identical | grep threads | tee $O/diag_different.txt
timeout 1500 python tools/soak_streams.py 0 3600 32 2>&1 | tee $O/soak_streams.txt
timeout 1500 python -m pytest tests/test_streams.py tests/test_glue_live.py tests/test_glue_app_patterns.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5 | tee $O/pytest_streams.txt
