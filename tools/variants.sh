#!/bin/bash
# Build kernel-variant libraries: tools/variants.sh name "<extra hipcc flags>" [name2 "<flags2>" ...] -> libde265_amd/variants/<name>.so
# (run the bench against one with M355_LIB=libde265_amd/variants/<name>.so python bench.py)
set -e
cd "$(dirname "$0")/../libde265_amd/csrc"
mkdir -p ../variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  d=/tmp/m355_variant_$name; rm -rf $d; mkdir -p $d
  for f in runtime runtime_upload runtime_decode runtime_shard runtime_ipc slots k_meta k_inter k_residual k_intra k_deblock k_sao k_shard k_hash; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -w -I../../include -I. $flags -c $f.hip -o $d/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../variants/$name.so $d/*.o
  echo built ../variants/$name.so
done
