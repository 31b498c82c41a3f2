#!/bin/bash
# Several rocprofv3 --pmc passes over bench.py (5 steps each) -> gpurun_out/<tag>/pmc_diag.txt (per-kernel means per dispatch)
TAG=${1:-diag}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; REPO=$PWD
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum" \
           "TA_BUSY_avr TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $OUT/p$i -o p --output-format csv -- python $REPO/bench.py --no-cpu-baseline --steps 4 --warmup 1 ${BENCH_ARGS} > $OUT/p$i.log 2>&1 || echo "pass $i failed: $set"
done
python $REPO/tools/pmc_summary.py $OUT/p* > $OUT/pmc_diag.txt 2>&1
find $OUT -name "*.csv" -size +5M -delete
python - <<PY
import sys
# transpose: per kernel, one counter per line
lines=open("$OUT/pmc_diag.txt").read().splitlines()
hdr=lines[0].split()
for l in lines[1:]:
    f=l.split()
    if not f: continue
    n=len(hdr)-2
    name=" ".join(f[:len(f)-n-1]); vals=f[len(f)-n:]
    if not any(k in name for k in ("k_inter","k_intra","k_sao","k_residual","k_deblock")): continue
    print("==", name)
    for h,v in zip(hdr[2:],vals):
        if v!="-": print("   %-24s %s"%(h,v))
PY
