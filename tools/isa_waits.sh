#!/bin/bash
# Which dependent memory round trips does a kernel REALLY have?  Compiles one .hip of libde265_amd/csrc for gfx950 to assembly and prints, for the
# kernels whose mangled name contains <pattern>, the sequence of vector / scalar memory instructions, waits, barriers and branches — the view that
# found the hidden trips of round 4 (vector loads from the kernel-argument segment, loads sunk below an early exit, a drain at a loop header, a wait
# per conditional load: DESIGN.md §4 Round 4, second half).
# usage: tools/isa_waits.sh <file.hip> <pattern> [extra hipcc flags]      e.g.  tools/isa_waits.sh k_sao.hip k_saoItLb1
set -e
F=$1; P=$2; shift 2
cd "$(dirname "$0")/../libde265_amd/csrc"
S=/tmp/isa_waits_$$.s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fvisibility=hidden -w -I../../include -I. "$@" -S --cuda-device-only -o $S $F
for k in $(grep -o "^_Z[A-Za-z0-9_]*$P[A-Za-z0-9_]*:" $S | tr -d ':' | sort -u); do
  echo "== $k"
  awk -v k="$k:" '$1==k{f=1} f{print} /^\.Lfunc_end/{if(f) exit}' $S > /tmp/isa_waits_$$.k
  echo "   $(grep -c '^\s*v_' /tmp/isa_waits_$$.k) VALU, $(grep -c '^\s*s_' /tmp/isa_waits_$$.k) SALU, $(grep -c -E '^\s*(global|flat|buffer)_load' /tmp/isa_waits_$$.k) vector loads ($(grep -c '^\s*flat_' /tmp/isa_waits_$$.k) flat), $(grep -c 'scratch_' /tmp/isa_waits_$$.k) scratch accesses (static counts)"
  grep -n -E "global_load|flat_load|buffer_load|scratch_|s_waitcnt vmcnt|s_barrier|s_load_dword|global_store|flat_store|Loop Header|s_endpgm" /tmp/isa_waits_$$.k | cut -c1-110
done
rm -f $S /tmp/isa_waits_$$.k
