import sys, ctypes; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from oracle_py import Oracle
from synth_util import device_decode, make_case, oracle_decode
from libde265_amd import capi
o = Oracle(ctypes.CDLL('oracle/liboracle.so'))
ctx = capi.Context(capi.Library(), 0)
for case in [dict(width=192, height=128, bit_depth=10, seed=13, tile_cols=3, tile_rows=1), dict(width=192, height=128, bit_depth=10, seed=13), dict(width=192, height=128, bit_depth=8, seed=13, tile_cols=3, tile_rows=1)]:
    pic, refs = make_case(**case)
    for st, name in [(1,"inter"),(3,"+res"),(7,"+intra"),(15,"+dbk"),(31,"+sao")]:
        for rep in range(2):
            g = device_decode(ctx, pic, refs, st); w = oracle_decode(o, pic, refs, st)
            bad = [int((a!=b).sum()) for a,b in zip(g,w)]
            print(case.get("tile_cols",1), case["bit_depth"], name, rep, bad)
