#!/bin/bash
# round 6, visit 25: the 128-register build with an LDS pad gave wrong 8-bit pictures (visit 24) — is there an out-of-bounds LDS read in the PRODUCT kernel?  padonly = the product's
# k_inter_jobs with 28 KB of garbage behind its LDS arrays and garbage in its EDGE slots; w4 / w4pad = the 128-register builds
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for v in padonly; do echo "== $v"; M355_LIB=$GRAFT_REPO_ROOT/libde265_amd/variants/$v.so timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_random.py tests/test_inter_extremes.py tests/test_inter_narrow.py -m gpu -q 2>&1 | tail -6; done
