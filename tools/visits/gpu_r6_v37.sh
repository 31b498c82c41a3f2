#!/bin/bash
# round 6, visit 37: damaged streams on the hardware (no crash, no unbounded wait; how many differ from the reference), concurrent decoders in one process,
# the sharded soak at size, big streams over three ranks
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v37; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
SOAK_DAMAGE=1 timeout 900 python tools/soak_streams.py 0 480 16 2>&1 | grep -v "^   (" | tail -8 | tee $O/soak_damaged.txt | cut -c1-300
echo "rc of the damaged soak: ${PIPESTATUS[0]}" | tee -a $O/soak_damaged.txt
timeout 600 python tools/soak_concurrent.py 5 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 2>&1 | tail -3 | tee $O/soak_concurrent.txt | cut -c1-300
SOAK_SCALE=6 timeout 900 python tools/soak_shard.py 50000 1600 16 2>&1 | tail -12 | tee $O/soak_shard_scale6.txt | cut -c1-300
M355_GLUE_RANKS=3 SOAK_BIG=4 timeout 900 python tools/soak_streams.py 12000 300 24 2>&1 | tail -8 | tee $O/soak_streams_big4_ranks3.txt | cut -c1-300
