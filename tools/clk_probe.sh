#!/bin/bash
# sample the shader clock while the C2 bench runs (is the dependency chain running at a low DPM state?)
mkdir -p gpurun_out/clk
rocm-smi --showperflevel --showclocks > gpurun_out/clk/idle.txt 2>&1
(timeout 120 python bench.py --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --workload c2_1080p_intra --steps 3000 --warmup 10 --pipeline-depth 1 --repeats 1 > gpurun_out/clk/bench.json 2>gpurun_out/clk/bench.err) &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" >> gpurun_out/clk/busy.txt; sleep 1; done
wait $BP
cat gpurun_out/clk/idle.txt | grep -i "level\|sclk\|mclk"; echo ---; cat gpurun_out/clk/busy.txt
# try forcing the high performance level, then measure again
rocm-smi --setperflevel high > gpurun_out/clk/set.txt 2>&1; tail -3 gpurun_out/clk/set.txt
timeout 120 python bench.py --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --workload c2_1080p_intra --steps 300 --warmup 10 --pipeline-depth 1 2>>gpurun_out/clk/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('perflevel high: %.4f ms/pic intra=%.4f' % (d['ms_per_step'], d['stage_ms']['intra']))"
rocm-smi --setperflevel auto >> gpurun_out/clk/set.txt 2>&1
