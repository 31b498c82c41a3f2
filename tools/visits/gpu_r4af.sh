#!/bin/bash
# round 4, visit af (ONE box, everything an A/B on it): k_residual / k_sao order and walk variants (tools/variants.sh: resxcd, respipe1, respipe1x,
# respipe2, saoxcd1, saoxcd2, allx), their fabric traffic (PMC FETCH_SIZE / WRITE_SIZE, product vs allx), the staging-arena ring (3 vs depth + 3),
# the sharded leg with 3 vs 4 pictures in flight
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4af; mkdir -p $O
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
run() { # name
  v=$1
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload c5_8k10_8tiles --steps 200 --warmup 10 --pipeline-depth 3 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s %.4f ms/pic (p10 %.4f p90 %.4f)  one-at-a-time %.4f  %s' % ('$v', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $O/variants.txt
  unset M355_LIB
}
for v in base resxcd respipe1 respipe1x respipe2 saoxcd2 allx base allx; do run $v; done
# the walk's grid
for g in 8192; do echo "M355_RES_PIPE_GRID=$g:" | tee -a $O/variants.txt; M355_RES_PIPE_GRID=$g run respipe1x; done
# the variants on the hardware parity suites (random pictures, synthetic configurations at full size)
for v in allx respipe2; do M355_LIB=$REPO/libde265_amd/variants/$v.so timeout 300 python -m pytest tests/test_gpu_random.py tests/test_gpu_synth.py -m gpu -x -q 2>&1 | tail -1 | sed "s/^/$v: /" | tee -a $O/variants_parity.txt; done
# staging-arena ring: 3 (rounds 1-3) vs depth + 3
for r in 3 0; do
  M355_TRANSIENT_RING=$r timeout 300 python bench.py --no-cpu-baseline --no-dependent-chain --no-end-to-end --steps 20 --warmup 5 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); u=d['with_upload']
print('M355_TRANSIENT_RING=$r (0 = depth + 3): with_upload %.4f ms  submit_only %.4f ms  copying %.4f ms  (resident lists %.4f ms)' % (u['ms_per_step'], u['submit_only']['ms_per_step'], u['copying_submit']['ms_per_step'], d['ms_per_step']))" | tee -a $O/ring.txt
done
M355_PROFILE_UPLOAD=1 timeout 120 python tools/prof_submit.py 2>&1 | tail -6 | tee $O/submit_host_phases.txt
# sharded leg, world 1: pictures in flight
for sd in 4 3; do
  M355_BENCH_SHARD_DEPTH=$sd timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$sd bench.py --gpus 1 --steps 100 --warmup 10 --force-tile-shard $B 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d.get('tile_sharded') or {}
print('sharded leg, %s in flight: unsharded %.4f ms, sharded %.4f ms (x%.3f), one at a time %.4f, host enqueue %.4f' % (t.get('pictures_in_flight'), d['ms_per_step'], t.get('ms_per_picture', 0), t.get('ms_per_picture', 0) / d['ms_per_step'], t.get('ms_per_picture_one_at_a_time', 0), t.get('host_enqueue_ms_per_picture', 0)))" | tee -a $O/sharded.txt
done
# fabric traffic of the variants' kernels: product vs allx (one picture in flight, 5 pictures)
cd /tmp; w=c5_8k10_8tiles
for v in base allx; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_rd_$v -o rd --output-format csv -- python $REPO/bench.py --workload $w --steps 5 --warmup 1 $B --pipeline-depth 1 > $O/pmc_rd_$v.log 2>&1
  python $REPO/tools/pmc_summary.py $O/pmc_rd_$v > $O/pmc_summary_$v.txt 2>&1; echo "== $v"; grep -E "residual|sao" $O/pmc_summary_$v.txt
done
unset M355_LIB
cd $REPO
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +5M -delete
