#!/bin/bash
# round 4, visit ah: what are the kernels of a C5 picture bound by?  SQ instruction counters per kernel (one picture in flight)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4ah; mkdir -p $O
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
cd /tmp; w=c5_8k10_8tiles
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $c -d $O/p$i -o x --output-format csv -- python $REPO/bench.py --workload $w --steps 4 --warmup 1 $B --pipeline-depth 1 > $O/p$i.log 2>&1 || tail -3 $O/p$i.log
done
python $REPO/tools/pmc_summary.py $O/p1 $O/p2 $O/p3 2>&1 | tee $O/sq_counters.txt | cut -c1-400 | head -30
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +5M -delete
