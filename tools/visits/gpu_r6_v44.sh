#!/bin/bash
# round 6, visit 44: the chain soak (tools/soak_chain.py) — as the race falls, with the chain schedules forced, and at size
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v44; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
timeout 900 python tools/soak_chain.py 0 6000 24 2>&1 | tail -12 | tee $O/soak_chain.txt | cut -c1-400
M355_TEST_CHAIN_LANES=1 M355_TEST_CHAIN_RESIDUALS=1 timeout 900 python tools/soak_chain.py 6000 6000 24 2>&1 | tail -12 | tee $O/soak_chain_forced.txt | cut -c1-400
SOAK_SCALE=8 timeout 900 python tools/soak_chain.py 12000 800 16 2>&1 | tail -12 | tee $O/soak_chain_scale8.txt | cut -c1-400
SOAK_SCALE=8 M355_TEST_CHAIN_LANES=1 timeout 900 python tools/soak_chain.py 13000 800 16 2>&1 | tail -12 | tee $O/soak_chain_scale8_forced.txt | cut -c1-400
