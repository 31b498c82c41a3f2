#!/usr/bin/env python3
"""Offline model of k_intra's block chain on an intra picture (host side only, no GPU): the per-CTB barrier levels as
runtime_upload.hip intra_schedule assigns them, the cross-CTB polls as k_intra makes them, block / barrier / hand-off times from
profiles/r03_*_intra_level_profile_* — to compare LEVEL POLICIES before building one:
    python tools/intra_sim.py [config] [u_us ovh_us handoff_us]
policies:  asap        = levels from in-CTB dependencies only (the shipped one), polls for every available halo entry
           asap+prune  = the same levels, halo entries the mode never reads are not polled
           timed       = levels from picture-wide earliest start times (cross-CTB arrivals included) + prune
"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from libde265_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "c2_1080p_intra"
U, OVH, HO = (float(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (0.40, 0.09, 2.5)
cfg = dict(synth.CONFIGS[name])
pic = synth.picture(**cfg)
pp = pic.pp[0]
L2C = int(pp["log2_ctb_size"]); W = int(pp["width"]); H = int(pp["height"])
ctbW = (W + (1 << L2C) - 1) >> L2C; ctbH = (H + (1 << L2C) - 1) >> L2C
cf = int(pp["chroma_format_idc"])
sw = 2 if cf in (1, 2) else 1; sh = 2 if cf == 1 else 1
flags = int(pp["flags"])
STRONG = bool(flags & 2); SMOOTH_OFF = bool(flags & 32); CIP = bool(flags & 1)
MAG = [0, 2, 5, 9, 13, 17, 21, 26, 32]


def used_entries(m, log2, cidx, ibflags):
    """border entries a mode reads per side (top, left), smoothing included; None = all (strong smoothing)"""
    nT = 1 << log2
    if cidx == 0 and log2 == 5 and STRONG: return 2 * nT, 2 * nT
    filt = False
    if not SMOOTH_OFF and (cidx == 0 or cf == 3) and m != 1 and log2 != 2:
        md = min(abs(m - 26), abs(m - 10))
        filt = md > 7 if log2 == 3 else (md > 1 if log2 == 4 else (md > 0 if log2 == 5 else False))
    bf = cidx == 0 and log2 < 5 and (m == 1 or not (ibflags & 2))
    if m == 0: te, le = nT + 1, nT + 1
    elif m == 1: te, le = nT, nT
    elif 10 < m < 26: te, le = nT, nT
    elif m == 26: te, le = nT, (nT if bf else 0)
    elif m == 10: le, te = nT, (nT if bf else 0)
    elif m > 26: te, le = min(2 * nT, nT + ((nT * MAG[m - 26]) >> 5) + 1), 0
    else: le, te = min(2 * nT, nT + ((nT * MAG[10 - m]) >> 5) + 1), 0
    if filt:
        if te: te = min(2 * nT, te + 1)
        if le: le = min(2 * nT, le + 1)
    return te, le


def sched_entries(m, log2, cidx, ibflags, ux, uy):
    """as intra_schedule (runtime_upload.hip): the conservative ranges the in-CTB levels are made from"""
    nT = 1 << log2
    if CIP or (cidx == 0 and log2 == 5 and STRONG): return 2 * nT, 2 * nT
    filt = False
    if not SMOOTH_OFF and (cidx == 0 or cf == 3) and m != 1 and log2 != 2:
        md = min(abs(m - 26), abs(m - 10))
        filt = md > 7 if log2 == 3 else (md > 1 if log2 == 4 else (md > 0 if log2 == 5 else False))
    bf = cidx == 0 and log2 < 5 and (m == 1 or not (ibflags & 2))
    te = le = 2 * nT
    if m == 0: te, le = nT + 1, 2 * nT
    elif m == 1: te, le = nT, nT
    elif 10 < m < 26: te, le = nT, nT
    elif m == 26: te, le = nT, (nT if (bf or uy == 0) else 0)
    elif m == 10: le, te = nT, (nT if (bf or ux == 0) else 0)
    elif m > 26: te, le = min(2 * nT, nT + ((nT * MAG[m - 26]) >> 5) + 2), (0 if uy > 0 else 2 * nT)
    else: le, te = 2 * nT, (0 if ux > 0 else 2 * nT)
    if filt:
        if te: te = min(2 * nT, te + 1)
        if le: le = 2 * nT
    return te, le


ibs = pic.ibs; ctbs = pic.ctbs
NB = len(ibs)
bx = ibs["x"].astype(int); by = ibs["y"].astype(int); bc = ibs["cidx"].astype(int); bl = ibs["log2_size"].astype(int)
bm = ibs["mode"].astype(int); bf_ = ibs["flags"].astype(int)
cost = np.where(bl >= 5, 4.5, np.where(bl == 4, 2.0, 1.0))
# per component: picture-wide grid of 4x4 units -> block index
gw = [(W + 3) // 4 + 40, ((W // sw) + 3) // 4 + 40, ((W // sw) + 3) // 4 + 40]
gh = [(H + 3) // 4 + 40, ((H // sh) + 3) // 4 + 40, ((H // sh) + 3) // 4 + 40]
grid = [np.full((gh[c], gw[c]), -1, np.int64) for c in range(3)]
b_ctb = np.zeros(NB, np.int64)
dep_in = [None] * NB      # in-CTB dependencies (scheduler ranges)
dep_x_all = [None] * NB   # cross-CTB producers polled today (every available entry)
dep_x_used = [None] * NB  # cross-CTB producers the mode reads
for c in range(ctbW * ctbH):
    cx, cy = c % ctbW, c // ctbW
    s, n = int(ctbs["ib_start"][c]), int(ctbs["ib_count"][c])
    for k in range(s, s + n):
        ci = bc[k]; csw = (sw == 2) if ci else 0; csh = (sh == 2) if ci else 0
        cu = (1 << L2C) >> (2 + csw); cv = (1 << L2C) >> (2 + csh)          # CTB size in units
        ux, uy = bx[k] >> 2, by[k] >> 2
        lux, luy = ux - cx * cu, uy - cy * cv
        n4 = (1 << bl[k]) >> 2
        b_ctb[k] = c
        g = grid[ci]
        din, dall, dused = set(), set(), set()
        if not (bf_[k] & 4):
            nT = 1 << bl[k]
            te_s, le_s = sched_entries(bm[k], bl[k], ci, bf_[k], lux, luy)
            te_u, le_u = used_entries(bm[k], bl[k], ci, bf_[k])
            tu_s, lu_s = (te_s + 3) >> 2, (le_s + 3) >> 2
            tu_u, lu_u = (te_u + 3) >> 2, (le_u + 3) >> 2
            for t in range(-1, 2 * n4):
                for (yy, xx, lim_s, lim_u) in ((uy + t, ux - 1, lu_s, lu_u), (uy - 1, ux + t, tu_s, tu_u)):
                    if yy < 0 or xx < 0: continue
                    d = g[yy, xx]
                    if d < 0: continue
                    same = b_ctb[d] == c
                    if same:
                        if t < lim_s: din.add(int(d))
                    else:
                        # a sample of another CTB: below the CTB row = not available (not registered yet: d < 0 covers it)
                        dall.add(int(d))
                        if t < lim_u or t == -1: dused.add(int(d))
        dep_in[k], dep_x_all[k], dep_x_used[k] = din, dall, dused
        g[uy:uy + n4, ux:ux + n4] = k

n_x_all = sum(len(d) for d in dep_x_all); n_x_used = sum(len(d) for d in dep_x_used)
print("%s: %d blocks, %d CTBs; cross-CTB producer edges: %d polled today, %d the modes read" % (name, NB, ctbW * ctbH, n_x_all, n_x_used))


def levels_asap():
    lv = np.zeros(NB, np.int64)
    for k in range(NB):
        lv[k] = max([lv[d] + 1 for d in dep_in[k]], default=0)
    return lv


def levels_timed(depx, width=1.0, ho=None):
    """picture-wide earliest start times in block units (infinite waves), cross-CTB edges cost a hand-off; per CTB the blocks
    sorted by that time, a new level where the time has advanced by `width` (or a dependency sits in the open level)"""
    ho = HO / U if ho is None else ho
    t = np.zeros(NB)
    for k in range(NB):
        a = max([t[d] + cost[d] for d in dep_in[k]], default=0.0)
        b = max([t[d] + cost[d] + ho for d in depx[k]], default=0.0)
        t[k] = max(a, b)
    lv = np.zeros(NB, np.int64)
    for c in range(ctbW * ctbH):
        s, n = int(ctbs["ib_start"][c]), int(ctbs["ib_count"][c])
        if not n: continue
        for ci in range(3):
            pass
        order = sorted(range(s, s + n), key=lambda k: (t[k], k))
        L = 0; t0 = t[order[0]]; cur = set()
        for k in order:
            if t[k] >= t0 + width or any(d in cur for d in dep_in[k]):
                L += 1; t0 = t[k]; cur = set()
            lv[k] = L; cur.add(k)
    return lv, t


GW = {"8+2+2": (8, 2, 2), "8+4+4": (8, 4, 4)}


def simulate(lv, depx, waves=(8, 2, 2), shared=0, u=U, ovh=OVH, ho=HO, prologue=6.0, pl=0.0):
    done = np.zeros(NB)
    total_levels = 0; stall = 0.0; busy = 0.0
    end_ctb = np.zeros(ctbW * ctbH)
    for c in range(ctbW * ctbH):
        s, n = int(ctbs["ib_start"][c]), int(ctbs["ib_count"][c])
        if not n: continue
        order = sorted(range(s, s + n), key=lambda k: (lv[k], bc[k], k))
        tcur = prologue
        i = 0
        while i < n:
            j = i
            while j < n and lv[order[j]] == lv[order[i]]: j += 1
            blocks = order[i:j]
            arrive = 0.0
            for k in blocks:
                for d in depx[k]: arrive = max(arrive, done[d] + ho)
            start = max(tcur, arrive)
            if pl and any(depx[k] for k in blocks): start = max(start, tcur + pl)     # a global round trip even when the granule is there
            stall += start - tcur
            load = {}
            cnt = [0, 0, 0]
            m = 0.0
            for k in blocks:
                ci = 0 if shared else bc[k]
                wv = (ci, cnt[ci] % (shared if shared else waves[ci])); cnt[ci] += 1
                load[wv] = load.get(wv, 0.0) + cost[k] * u
                done[k] = start + load[wv]
                m = max(m, load[wv])
            tcur = start + m + ovh
            busy += m + ovh
            total_levels += 1
            i = j
        end_ctb[c] = tcur
    return end_ctb.max(), total_levels, stall, busy


def levels_greedy(depx, waves=(8, 2, 2), u=U, ovh=OVH, ho=HO, prologue=6.0, width=0.0):
    """levels formed BY the model: at every step of a CTB the blocks whose in-CTB dependencies are done and whose cross-CTB inputs
    have (by the model's clock) arrived make the next level; when nothing has arrived the CTB's clock jumps to the first arrival"""
    lv = np.zeros(NB, np.int64)
    done = np.zeros(NB)
    users = [[] for _ in range(NB)]
    for k in range(NB):
        for d in dep_in[k]: users[d].append(k)
    for c in range(ctbW * ctbH):
        s, n = int(ctbs["ib_start"][c]), int(ctbs["ib_count"][c])
        if not n: continue
        indeg = {k: len(dep_in[k]) for k in range(s, s + n)}
        arr = {k: max([done[d] + ho for d in depx[k]], default=0.0) for k in range(s, s + n)}
        ready = sorted(k for k in indeg if indeg[k] == 0)
        tcur = prologue; L = 0; left = n
        while left:
            cand = [k for k in ready if arr[k] <= tcur + 1e-9]
            if not cand:
                tcur = min(arr[k] for k in ready)
                cand = [k for k in ready if arr[k] <= tcur + width]
            cand.sort(key=lambda k: (bc[k], k))
            load = {}; cnt = [0, 0, 0]; m = 0.0
            for k in cand:
                ci = bc[k]; wv = (ci, cnt[ci] % waves[ci]); cnt[ci] += 1
                load[wv] = load.get(wv, 0.0) + cost[k] * u
                done[k] = tcur + load[wv]; m = max(m, load[wv]); lv[k] = L
            tcur += m + ovh; L += 1; left -= len(cand)
            cs = set(cand)
            ready = [k for k in ready if k not in cs]
            for k in cand:
                for q in users[k]:
                    indeg[q] -= 1
                    if indeg[q] == 0: ready.append(q)
    return lv


lv0 = levels_asap()
for label, lv, dx in (("asap (shipped)", lv0, dep_x_all), ("asap + pruned polls", lv0, dep_x_used)):
    T, nl, st, bs = simulate(lv, dx)
    print("%-34s %7.1f us  %6d levels  CTB-time stalled %8.0f us, working %8.0f us" % (label, T, nl, st, bs))
for width in (0.5, 1.0, 2.0, 3.0):
    lv1, t1 = levels_timed(dep_x_used, width)
    T, nl, st, bs = simulate(lv1, dep_x_used)
    print("%-34s %7.1f us  %6d levels  CTB-time stalled %8.0f us, working %8.0f us   (ideal chain %.0f units)" % ("timed width %.1f + pruned" % width, T, nl, st, bs, (t1 + cost).max()))
    T, nl, st, bs = simulate(lv1, dep_x_used, waves=(8, 4, 4))
    print("%-34s %7.1f us" % ("   ... with 8+4+4 waves", T))
for width in (0.0, 0.3, 1.0):
    lv2 = levels_greedy(dep_x_used, width=width)
    for lab, kw in (("model's own clock", {}), ("blocks 20 %% slower than assumed", dict(u=U * 1.2)), ("hand-off 4 us, not %.1f" % HO, dict(ho=4.0)), ("hand-off 1.5 us", dict(ho=1.5))):
        T, nl, st, bs = simulate(lv2, dep_x_used, **kw)
        print("%-34s %7.1f us  %6d levels  stalled %8.0f us, working %8.0f us   [%s]" % ("greedy (width %.1f) + pruned" % width, T, nl, st, bs, lab))
for lab, kw in (("blocks 20 % slower", dict(u=U * 1.2)), ("hand-off 4 us", dict(ho=4.0)), ("hand-off 1.5 us", dict(ho=1.5))):
    T, nl, st, bs = simulate(lv0, dep_x_all, **kw)
    print("%-34s %7.1f us  [%s]" % ("asap (shipped)", T, lab))
# composition of the ideal critical chain (infinite waves): walk back from the last block
ho_u = HO / U
t = np.zeros(NB); pred = np.full(NB, -1, np.int64); cross = np.zeros(NB, bool)
for k in range(NB):
    best, bp, bx_ = 0.0, -1, False
    for d in dep_in[k]:
        if t[d] + cost[d] > best: best, bp, bx_ = t[d] + cost[d], d, False
    for d in dep_x_used[k]:
        if t[d] + cost[d] + ho_u > best: best, bp, bx_ = t[d] + cost[d] + ho_u, d, True
    t[k], pred[k], cross[k] = best, bp, bx_
k = int(np.argmax(t + cost)); n_by = {2: 0, 3: 0, 4: 0, 5: 0}; n_cross = 0; n_comp = [0, 0, 0]
while k >= 0:
    n_by[int(bl[k])] += 1; n_comp[bc[k]] += 1
    if cross[k]: n_cross += 1
    k = int(pred[k])
print("ideal chain: %d blocks (4x4 %d, 8x8 %d, 16x16 %d, 32x32 %d; luma %d cb %d cr %d), %d hand-offs = %.0f of %.0f units" % (sum(n_by.values()), n_by[2], n_by[3], n_by[4], n_by[5], n_comp[0], n_comp[1], n_comp[2], n_cross, n_cross * ho_u, (t + cost).max()))

print("--- with a poll round trip for every level that reads another CTB's samples (the halo is staged before the neighbours finish)")
nlv = 0
for c in range(ctbW * ctbH):
    s_, n_ = int(ctbs["ib_start"][c]), int(ctbs["ib_count"][c])
    nlv += len(set(int(lv0[k]) for k in range(s_, s_ + n_) if dep_x_all[k]))
print("shipped: %d of %d levels hold a block that reads another CTB" % (nlv, 9837))
lvg = levels_greedy(dep_x_used)
for pl in (0.5, 1.0, 1.5):
    a = simulate(lv0, dep_x_all, pl=pl)[0]; b = simulate(lv0, dep_x_used, pl=pl)[0]; c_ = simulate(lvg, dep_x_used, pl=pl)[0]; d_ = simulate(lvg, dep_x_used, pl=0)[0]
    print("poll %.1f us: shipped %.0f us, pruned %.0f, greedy + pruned %.0f, greedy + pruned + halo kept fresh by a poller wave %.0f" % (pl, a, b, c_, d_))


def levels_timed_int(depx, ho_u=6, c32=5, c16=2):
    """integer clock: block costs 1 / c16 / c32 units, a hand-off ho_u; a CTB's levels = the distinct start times of its blocks"""
    ci_ = np.where(bl >= 5, c32, np.where(bl == 4, c16, 1)).astype(np.int64)
    t = np.zeros(NB, np.int64)
    for k in range(NB):
        a = max([t[d] + ci_[d] for d in dep_in[k]], default=0)
        b = max([t[d] + ci_[d] + ho_u for d in depx[k]], default=0)
        t[k] = max(a, b)
    lv = np.zeros(NB, np.int64)
    for c in range(ctbW * ctbH):
        s, n = int(ctbs["ib_start"][c]), int(ctbs["ib_count"][c])
        if not n: continue
        ts = sorted(set(int(t[k]) for k in range(s, s + n)))
        rank = {v: i for i, v in enumerate(ts)}
        for k in range(s, s + n): lv[k] = rank[int(t[k])]
    return lv


print("--- integer-clock levels (what the host can compute in one pass over the CTBs in wavefront order)")
for ho_u, c32, c16 in ((6, 5, 2), (6, 4, 2), (8, 5, 2), (4, 5, 2), (10, 5, 2)):
    lv3 = levels_timed_int(dep_x_used, ho_u, c32, c16)
    res = []
    for kw in ({}, dict(pl=0.5), dict(u=U * 1.2), dict(ho=4.0), dict(ho=1.5)):
        T, nl, st, bs = simulate(lv3, dep_x_used, **kw); res.append(T)
    print("hand-off %2d units, 32x32 %d, 16x16 %d: %6d levels; model %.0f us | +0.5 us poll floor %.0f | blocks 20 %% slower %.0f | hand-off 4 us %.0f | 1.5 us %.0f" % (ho_u, c32, c16, nl, *res))
