for p in 1 0; do
export M355_LANE_PRIORITIES=$p
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 50 --warmup 5 --force-tile-shard --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['tile_sharded']; print('prio=$p unsharded %.4f sharded in flight %.4f (%d) one %.4f enqueue %.4f nonref %.4f' % (d['ms_per_step'], t['ms_per_picture'], t['pictures_in_flight'], t['ms_per_picture_one_at_a_time'], t['host_enqueue_ms_per_picture'], t['non_reference_picture']['ms_per_picture']))"
for d in 3 4 6; do timeout 300 python bench.py --workload c5_8k10_8tiles --steps 100 --warmup 10 --pipeline-depth $d --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio=$p c5 depth $d: %.4f ms/pic' % d['ms_per_step'])"; done
done
timeout 300 python bench.py --workload c2_1080p_intra --steps 100 --warmup 10 --no-cpu-baseline --no-dependent-chain --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 with_upload', json.dumps(d['with_upload'])[:400])"
GPU_MAX_HW_QUEUES=16 timeout 300 python bench.py --workload c2_1080p_intra --steps 200 --warmup 10 --pipeline-depth 8 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 hwq16 depth 8: %.4f ms/pic' % d['ms_per_step'])"
