#!/usr/bin/env python3
"""Soak of the product kernels under the SIMT interpreter against the oracle on random pictures beyond the CPU tier's selection (the seeds of
tests/test_gpu_random.py the tier skips + 1000..1299); meant for switches: M355_FUSE_DBH=1 M355_INTRA_ONE_SIDED=1 python tools/soak_emu.py"""
import sys, ctypes; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from libde265_amd import capi
from oracle_py import Oracle
from synth_util import assert_planes_equal, device_decode, make_case, oracle_decode
from test_gpu_random import random_case
lib = capi.Library('/root/repo/tests/simt_emu/_build/libde265_mi355x_emu.so')
o = Oracle(ctypes.CDLL('/root/repo/oracle/liboracle.so'))
n = 0
for seed in list(range(200)) + list(range(1000, 1300)):
    if seed < 200 and seed % 3 == 0: continue
    pic, refs = make_case(**random_case(seed))
    ctx = capi.Context(lib, 0)
    try:
        assert_planes_equal(device_decode(ctx, pic, refs), oracle_decode(o, pic, refs), "seed %d" % seed)
    finally:
        ctx.close()
    n += 1
print("soak ok:", n, "pictures")
