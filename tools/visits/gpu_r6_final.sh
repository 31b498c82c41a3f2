#!/bin/bash
# round 6, the final tree (per-lane k_inter with VOP3P heads, interprocess transport, forced-schedule tests, frame-checked bench legs): whole GPU tier, the driver's bench command, one JSON per BASELINE config (with cpu_baseline), kernel traces,
# PMC traffic passes (-> profiles/pmc_traffic.json) and SQ counters of C5, the N > 1 path at world size 1
#   gpurun --timeout 2400 -- 'bash tools/visits/gpu_r6_final.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6zz; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "GPU tier"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
stamp "driver's command"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench.err ) 2>&1 | grep real | tee -a $O/timeline.txt
python -c "
import json; d=json.loads(open('$O/bench_driver_line.json').read().strip().splitlines()[-1]); print('driver-like:', d['value'], d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'])" | tee -a $O/timeline.txt
stamp "other configs"
for w in c2_1080p_intra c3_4k_inter c4_4k_4tiles; do E=--no-end-to-end; [ $w = c3_4k_inter ] && E=; timeout 600 python bench.py --workload $w --steps 100 --warmup 10 $E > $O/bench_$w.json 2>>$O/bench.err; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['ms_per_step_one_in_flight'], d['stage_ms'])" | tee -a $O/timeline.txt; done
timeout 300 python bench.py --workload c2_1080p_intra --steps 64 --warmup 8 --pipeline-depth 32 --intra-batch 8 --no-end-to-end --no-cpu-baseline > $O/bench_c2_batch.json 2>>$O/bench.err
timeout 300 python bench.py --workload c2_1080p_intra --steps 100 --warmup 10 --pipeline-depth 1 --no-end-to-end --no-cpu-baseline --no-with-upload --no-dependent-chain --no-cold-refs > $O/bench_c2_depth1.json 2>>$O/bench.err
for f in c2_batch c2_depth1; do python -c "
import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_one_in_flight'], d['stage_ms'])" | tee -a $O/timeline.txt; done
python -c "
import json
for w in ('driver_line','c3_4k_inter','c4_4k_4tiles'):
    d=json.loads(open('$O/bench_%s.json' % w).read().strip().splitlines()[-1])
    print(w, 'chain', json.dumps(d.get('dependent_chain'))[:200]); print(w, 'rotating', json.dumps(d.get('rotating_references'))[:200]); print(w, 'with_upload', json.dumps(d.get('with_upload'))[:300]); print(w, 'with_transfers', json.dumps(d.get('with_transfers'))[:300]); print(w, 'e2e', json.dumps(d.get('end_to_end'))[:400]); print(w, 'cpu', json.dumps(d.get('cpu_baseline'))[:300]); print(w, 'roofline', json.dumps(d.get('roofline'))[:400])" | tee -a $O/timeline.txt
stamp "kernel traces"
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --no-cold-refs"
cd /tmp
for w in c5_8k10_8tiles c3_4k_inter c2_1080p_intra; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -o x --output-format csv -- python $REPO/bench.py $B --workload $w --steps 50 --warmup 5 > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -24 > $O/${w}_kernel_stats.txt
done
stamp "PMC traffic + SQ counters (C5, one picture in flight)"
w=c5_8k10_8tiles
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_rd -o rd --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 $B --pipeline-depth 1 > $O/pmc_rd.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_wr -o wr --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 $B --pipeline-depth 1 > $O/pmc_wr.log 2>&1
python $REPO/tools/pmc_summary.py $O/pmc_rd $O/pmc_wr > $O/pmc_summary_c5.txt 2>&1
python $REPO/tools/pmc_traffic.py $w "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of round 6, final tree (profiles/r06_zz_c5_8k10_8tiles_pmc_summary.txt)" $O/pmc_rd $O/pmc_wr > /dev/null
w=c3_4k_inter
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_rd3 -o rd --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 $B --pipeline-depth 1 > $O/pmc_rd3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_wr3 -o wr --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 $B --pipeline-depth 1 > $O/pmc_wr3.log 2>&1
python $REPO/tools/pmc_traffic.py $w "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of round 6, final tree" $O/pmc_rd3 $O/pmc_wr3 > /dev/null
cp $REPO/profiles/pmc_traffic.json $O/pmc_traffic.json
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c -d $O/p$i -o x --output-format csv -- python $REPO/bench.py --workload c5_8k10_8tiles --steps 4 --warmup 1 $B --pipeline-depth 1 > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
python $REPO/tools/pmc_summary.py $O/p1 $O/p2 $O/p3 2>&1 | cut -c1-400 | head -30 > $O/sq_counters_c5.txt
cd $REPO
stamp "world size 1 through the launcher (tile-sharded leg forced)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 100 --warmup 10 --force-tile-shard $B > $O/bench_world1.json 2>> $O/bench.err
python -c "
import json; d=json.loads(open('$O/bench_world1.json').read().strip().splitlines()[-1]); print('world 1: unsharded', d['ms_per_step'], 'sharded', json.dumps(d.get('tile_sharded'))[:500])" | tee -a $O/timeline.txt
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +5M -delete; find $O -name "*kernel_trace.csv" -size +5M -delete
stamp done
