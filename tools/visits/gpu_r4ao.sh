#!/bin/bash
# round 4, visit ao (final tree): PMC traffic passes of C5 (-> profiles/pmc_traffic.json), the driver's bench command again, the other
# BASELINE configurations, the world-1 sharded leg, quick hardware parity of the last change
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4ao; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_random.py tests/test_gpu_pipeline.py tests/test_gpu_girlshy.py -m gpu -x -q 2>&1 | tail -1 | tee $O/parity.txt
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
cd /tmp; w=c5_8k10_8tiles
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_rd -o rd --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 $B --pipeline-depth 1 > $O/pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_wr -o wr --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 $B --pipeline-depth 1 > $O/pmc_wr.log 2>&1
python $REPO/tools/pmc_summary.py $O/pmc_rd $O/pmc_wr > $O/pmc_summary_c5.txt 2>&1
python $REPO/tools/pmc_traffic.py $w "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of round 4, final tree (profiles/r04_ao_c5_8k10_8tiles_pmc_summary.txt)" $O/pmc_rd $O/pmc_wr > /dev/null
cp $REPO/profiles/pmc_traffic.json $O/pmc_traffic.json
cd $REPO
( time timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('driver-like:', d['value'], d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'], 'traffic', d['roofline']['traffic']['bytes_per_launch'], d['roofline']['traffic_total']['bytes_per_picture'], 'submit_only', d['with_upload']['submit_only']['ms_per_step'], 'with_upload', d['with_upload']['ms_per_step'], 'e2e', d['end_to_end'].get('speedup'), d['end_to_end']['with_output'].get('speedup'))"
for w in c2_1080p_intra c3_4k_inter c4_4k_4tiles; do timeout 300 python bench.py --workload $w --steps 100 --warmup 10 --no-end-to-end > $O/bench_$w.json 2>>$O/bench.err; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['ms_per_step_one_in_flight'], d['stage_ms'])"; done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 100 --warmup 10 --force-tile-shard $B > $O/bench_world1.json 2>> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_world1.json').read().strip().splitlines()[-1]); print('world 1: unsharded', d['ms_per_step'], 'sharded', json.dumps(d.get('tile_sharded'))[:600])"
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +5M -delete
