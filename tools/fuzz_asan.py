#!/usr/bin/env python3
"""Corrupted work lists through the product kernels compiled under the SIMT interpreter WITH AddressSanitizer: any
out-of-bounds access a validated-but-corrupt list provokes in a kernel is reported.  Build + run:
    make -C libde265_amd/csrc emu EMUOUT=/tmp/emu_asan EMUFLAGS="-O1 -g -std=c++17 -fPIC -x c++ -I../../tests/simt_emu -I../../include -I. -w -fsanitize=address -fno-omit-frame-pointer"
    g++ -shared -fsanitize=address -o /tmp/emu_asan/libde265_mi355x_emu.so /tmp/emu_asan/*.o
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 python tools/fuzz_asan.py
(last runs, round 4's last tree: 480 corrupted pictures, 197 accepted by validation and decoded, 283 rejected, no ASan report — as built, and once
more with the prepared switches on: M355_DEVICE_WORKLIST=1 M355_MERGE_TU_PLAN=1 M355_CLEAR_IN_COUNT_MIN=1; the round's last session: the same counts with
M355_FUSE_DBH=1 M355_INTRA_ONE_SIDED=1; round 5's last tree: the same counts as built and with M355_TEST_CHAIN_RESIDUALS=1 M355_CLEAR_IN_COUNT_MIN=1, plus — same build — the chain
bookkeeping of tests/test_emu_chain.py under M355_TEST_CHAIN_LANES=1 and the 29 cases of tests/test_emu_synth.py: no report)"""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from libde265_amd import capi, worklist
from synth_util import make_case
from test_emu_fuzz import corrupt
lib = capi.Library('/tmp/emu_asan/libde265_mi355x_emu.so')
ctx = capi.Context(lib, 0)
acc=rej=0
for seed in range(6):
    rng = np.random.default_rng(2000+seed)
    cfg = [dict(width=128, height=64, bit_depth=8, seed=301, tile_cols=2), dict(width=96, height=96, bit_depth=10, seed=302, intra_pct=60, features=31)][seed%2]
    for it in range(80):
        pic, refs = make_case(**cfg); pp = pic.pp[0]
        what=[corrupt(pic, rng) for _ in range(1+rng.integers(3))]
        hs=[ctx.frame_create_for(pp) for _ in refs]
        for h,pl in zip(hs,refs): ctx.frame_upload(h,pl)
        pic.dst_frame=ctx.frame_create_for(pp); pic.ref_frames=[hs[i] if i<len(hs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
        try:
            ctx.submit(pic); ctx.wait(); acc+=1
        except capi.M355Error as e:
            rej+=1
        ctx.frame_destroy(pic.dst_frame)
        for h in hs: ctx.frame_destroy(h)
print("accepted",acc,"rejected",rej)
