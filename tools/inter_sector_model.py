#!/usr/bin/env python3
"""What k_inter_jobs MUST move for a given synthetic picture, at the granularity the memory system works in.

For every PB and list the reference window ((w+7) x (h+7) luma, (w/2+3) x (h/2+3) per chroma plane, coordinates clamped to
the plane like motion.cc:141-159) is marked in per-(reference, plane) bitmaps at three granularities: samples (distinct
bytes), 64-byte sectors (what one L2 miss fetches from the fabric at least) and 128-byte L2 lines.  The sums are lower
bounds of the kernel's fabric READ traffic under a perfect cache (every sector fetched exactly once per launch) — compare
with the PMC figure in profiles/pmc_traffic.json.  The same is done for the prediction WRITES (always whole samples of the
destination; the bound there is the written area rounded to sectors).

    python tools/inter_sector_model.py [--workload c5] [--pitch-align 256]

CPU only (numpy); used for the DESIGN.md note on what bounds k_inter_jobs.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libde265_amd import synth, worklist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c5_8k10_8tiles")
    ap.add_argument("--pitch-align", type=int, default=256, help="row pitch alignment of a frame plane in bytes")
    args = ap.parse_args()
    cfg = dict(synth.CONFIGS[args.workload])
    pic = synth.picture(**cfg)
    pp = pic.pp[0]
    W, H = int(pp["width"]), int(pp["height"])
    cf = int(pp["chroma_format_idc"])
    B = 1 if pp["bit_depth_luma"] <= 8 else 2
    dims = worklist.plane_dims(W, H, cf)
    pbs = pic.pbs
    n = len(pbs)
    maps = {}   # (slot, plane) -> bool [h][w] sample map

    def smap(slot, c):
        k = (slot, c)
        if k not in maps:
            maps[k] = np.zeros((dims[c][1], dims[c][0]), dtype=bool)
        return maps[k]

    alg = 0
    x_, y_, w_, h_, fl = pbs["x"], pbs["y"], pbs["w"], pbs["h"], pbs["flags"]
    mv, rs = pbs["mv"], pbs["ref_slot"]
    for i in range(n):
        for l in range(2):
            if not (fl[i] & (worklist.PBF_MC_L0 << l)):
                continue
            if fl[i] & (worklist.PBF_FILL_L0 << l):
                continue
            mvx, mvy = int(mv[i][l][0]), int(mv[i][l][1])
            for c in range(3 if cf else 1):
                pw, ph = dims[c]
                if c == 0:
                    x0, y0, w, h = int(x_[i]) + (mvx >> 2) - 3, int(y_[i]) + (mvy >> 2) - 3, int(w_[i]) + 7, int(h_[i]) + 7
                else:
                    x0, y0 = (int(x_[i]) >> 1) + (mvx >> 3) - 1, (int(y_[i]) >> 1) + (mvy >> 3) - 1
                    w, h = (int(w_[i]) >> 1) + 3, (int(h_[i]) >> 1) + 3
                alg += w * h * B
                xa, xb = min(max(x0, 0), pw - 1), min(max(x0 + w - 1, 0), pw - 1)
                ya, yb = min(max(y0, 0), ph - 1), min(max(y0 + h - 1, 0), ph - 1)
                smap(int(rs[i][l]), c)[ya:yb + 1, xa:xb + 1] = True

    def granules(m, g):
        """bytes covered when the sample map is rounded to g-byte granules along a row (pitch is a multiple of g)"""
        per = g // B
        hh, ww = m.shape
        pad = (-ww) % per
        mm = np.pad(m, ((0, 0), (0, pad))) if pad else m
        return int(mm.reshape(hh, -1, per).any(axis=2).sum()) * g

    rd = {"algorithmic": alg, "distinct": 0, "sector64": 0, "line128": 0}
    for (slot, c), m in maps.items():
        rd["distinct"] += int(m.sum()) * B
        rd["sector64"] += granules(m, 64)
        rd["line128"] += granules(m, 128)

    # writes: the union of the PB areas per plane
    wr = {"algorithmic": 0, "sector64": 0, "line128": 0}
    for c in range(3 if cf else 1):
        m = np.zeros((dims[c][1], dims[c][0]), dtype=bool)
        s = 0 if c == 0 else 1
        for i in range(n):
            m[int(y_[i]) >> s:(int(y_[i]) + int(h_[i])) >> s, int(x_[i]) >> s:(int(x_[i]) + int(w_[i])) >> s] = True
        wr["algorithmic"] += int(m.sum()) * B
        wr["sector64"] += granules(m, 64)
        wr["line128"] += granules(m, 128)
    wr["pb_of"] = int((w_.astype(np.int64) * h_.astype(np.int64)).sum()) // 16 * 4
    out = {"workload": args.workload, "n_pbs": n, "bytes_per_sample": B, "read_MB": {k: round(v / 1e6, 1) for k, v in rd.items()},
           "write_MB": {k: round(v / 1e6, 1) for k, v in wr.items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
