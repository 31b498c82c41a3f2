tools/bench_variants.sh base noil base noil
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/il_rd -o rd --output-format csv -- python $R/bench.py --no-cpu-baseline --steps 5 --warmup 1 > /dev/null 2>&1
python $R/tools/pmc_summary.py $R/gpurun_out/il_rd | grep -E "kernel|k_inter"
