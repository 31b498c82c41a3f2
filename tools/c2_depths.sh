for d in 3 4 5 6 8; do timeout 300 python bench.py --workload c2_1080p_intra --steps 200 --warmup 10 --pipeline-depth $d --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 depth $d (M355_LANE_PRIORITIES=${M355_LANE_PRIORITIES:-unset}): %.4f ms/pic = %.3f M CTB64/s (one at a time %.4f)' % (d['ms_per_step'], 510/d['ms_per_step']/1e3, d['ms_per_step_one_in_flight']))"; done
