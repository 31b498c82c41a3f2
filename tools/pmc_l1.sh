#!/bin/bash
# L1 / texture-addresser counters of k_inter_jobs for kernel variants.  (The TA_*_STALLED_BY_* counters are left out: that pass
# never finished on this pool — rocprofv3 aborted after its 300 s limit with an incomplete dispatch.)  usage: tools/pmc_l1.sh base|<variant> ...
REPO=$PWD; export TMPDIR=/tmp
FL="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
mkdir -p $REPO/gpurun_out/pmc_l1
if [ ! -f $REPO/gpurun_out/pmc_l1/avail.txt ]; then rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TD|SQ|TCC)_[A-Z0-9_a-z]+" | sort -u > $REPO/gpurun_out/pmc_l1/avail.txt; fi
for v in "$@"; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  OUT=$REPO/gpurun_out/pmc_l1/$v; rm -rf $OUT; mkdir -p $OUT
  cd /tmp
  for c in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" "TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-60)
    timeout 300 rocprofv3 --pmc $c -d $OUT/$n -o x --output-format csv -- python $REPO/bench.py $FL --steps 4 --warmup 1 --pipeline-depth 1 > $OUT/$n.log 2>&1 || tail -2 $OUT/$n.log
  done
  python $REPO/tools/pmc_summary.py $OUT 2>&1 | grep -E "^kernel|k_inter" | tee $REPO/gpurun_out/pmc_l1/$v.txt
  rm -rf $OUT
done
