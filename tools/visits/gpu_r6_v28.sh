#!/bin/bash
# round 6, visit 28: the bitstream-level soak on the hardware (tools/soak_streams.py: 600 random generated streams through the reference and through the glue on the
# product backend, 16 processes sharing the GPU), then the whole GPU tier and the driver's bench command on the tree as it stands
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v28; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
timeout 1200 python tools/soak_streams.py 0 600 16 2>&1 | tee $O/soak_streams.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_all.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench.err
tail -c 600 $O/bench_driver_line.json
