#!/bin/bash
# round 5, visit 2: the lean k_inter_jobs (parity, then variants on ONE box), the two micro-benchmarks the review asked for
# (VALU issue rates, inverse DCT on the matrix pipe), the new defaults (one-sided intra levels, merged TU + plan launch) under the
# whole GPU tier, the device work list's host phases.
#   gpurun --timeout 2400 -- 'bash tools/gpu_r5_v2.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v2; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"

stamp "micro-benchmarks"
timeout 300 tools/ubench/_build/ub_valu_rate > $O/ub_valu_rate.txt 2>&1; tail -3 $O/ub_valu_rate.txt
timeout 300 tools/ubench/_build/ub_mfma_idct > $O/ub_mfma_idct.txt 2>&1; cat $O/ub_mfma_idct.txt
(cd /tmp; timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma_pmc -o x --output-format csv -- $REPO/tools/ubench/_build/ub_mfma_idct 4096 > $O/mfma_pmc.log 2>&1; python $REPO/tools/pmc_summary.py $O/mfma_pmc 2>&1 | cut -c1-200 | tee $O/ub_mfma_idct_pmc.txt)

stamp "whole GPU tier (lean k_inter_jobs, one-sided intra levels and merged TU + plan as defaults)"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt

stamp "k_inter_jobs variants, same box"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2; do for v in nolean lean_p0w3 lean_p1w3 base lean_p2w4 lean_p1w5; do for wd in "c5_8k10_8tiles 3" "c5_8k10_8tiles 1" "c3_4k_inter 3" "c4_4k_4tiles 3"; do set -- $wd
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line $v $1 $2 | tee -a $O/inter_variants.txt
done; done; done
unset M355_LIB

stamp "kernel trace + SQ counters of the new default (C5 one picture in flight; C3)"
cd /tmp
for w in c5_8k10_8tiles c3_4k_inter; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -o x --output-format csv -- python $REPO/bench.py $B --workload $w --steps 50 --warmup 5 > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -24 > $O/${w}_kernel_stats.txt
done
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c -d $O/p$i -o x --output-format csv -- python $REPO/bench.py --workload c5_8k10_8tiles --steps 4 --warmup 1 $B --pipeline-depth 1 > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
python $REPO/tools/pmc_summary.py $O/p1 $O/p2 $O/p3 2>&1 | cut -c1-400 | head -30 > $O/sq_counters_c5.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c -d $O/t_$c -o x --output-format csv -- python $REPO/bench.py --workload c5_8k10_8tiles --steps 10 --warmup 2 $B --pipeline-depth 1 > $O/t_$c.log 2>&1 || tail -3 $O/t_$c.log
done
python $REPO/tools/pmc_summary.py $O/t_FETCH_SIZE $O/t_WRITE_SIZE 2>&1 | cut -c1-200 | head -30 > $O/pmc_traffic_c5.txt
cd $REPO

stamp "device work list: host phases of the submit, both settings, 3 times"
for rep in 1 2 3; do for m in 0 1; do
  M355_DEVICE_WORKLIST=$m timeout 120 python tools/prof_submit.py 2>/dev/null | tail -1 | sed "s/^/DEVICE_WORKLIST=$m: /" | tee -a $O/worklist_submit.txt
done; done
for m in 0 1; do M355_DEVICE_WORKLIST=$m M355_PROFILE_UPLOAD=1 timeout 120 python tools/prof_submit.py 2>&1 | grep -v "^per step" | tail -4 | sed "s/^/DEVICE_WORKLIST=$m: /" | tee -a $O/worklist_submit.txt; done

stamp "driver's line on this box"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2>>$O/bench.err
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +5M -delete; find $O -name "*kernel_trace.csv" -size +5M -delete
stamp done
