#!/usr/bin/env python3
"""Diagnose tile-sharded decoding on the GPU: per stage mask and rank count, where does the sharded result differ from the oracle?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
from libde265_amd import capi, synth, worklist
from oracle_py import Oracle
from shard_util import local_sharded_decode
from synth_util import make_case, oracle_decode

name = sys.argv[1] if len(sys.argv) > 1 else "c5_8k10_8tiles"
lib = capi.Library()
o = Oracle(ctypes.CDLL(os.path.join(ROOT, "oracle", "liboracle.so")))
pic, refs = make_case(**synth.CONFIGS[name])
W = worklist
for st, label in ((W.STAGE_INTER, "inter"), (W.STAGE_INTER | W.STAGE_RESIDUAL, "inter+res"), (W.STAGE_INTER | W.STAGE_RESIDUAL | W.STAGE_INTRA, "recon"),
                  (W.STAGE_ALL & ~W.STAGE_SAO, "nosao"), (W.STAGE_ALL, "all")):
    want = oracle_decode(o, pic, refs, st)
    for n in [int(a) for a in sys.argv[2:]] or [2, 8]:
        got = local_sharded_decode(lib, pic, refs, n, device="cuda:0", stages=st)
        for r, g in enumerate(got):
            bad = [(c, int((a != b).sum()), np.argwhere(a != b).min(0).tolist(), np.argwhere(a != b).max(0).tolist()) for c, (a, b) in enumerate(zip(g, want)) if (a != b).any()]
            if bad:
                print(label, "ranks", n, "rank", r, bad, flush=True)
                break
        else:
            print(label, "ranks", n, "OK", flush=True)
