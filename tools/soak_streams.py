#!/usr/bin/env python3
"""Soak at the BITSTREAM level: random streams from oracle/_ref/streamgen (our coding-tree writer on the reference's own CABAC encoder: size, bit depth, tiles, slices,
chroma format, B pictures / random-access groups, SAO, and a random subset of the feature and geometry bits of tests/test_streams.py) decoded by the REFERENCE library
(oracle/_ref/libde265_ref.so = the checker) and by glue/_build/libde265.so on the product backend; the MD5 of the output and the number of pictures must agree and no CPU
pixel kernel may run.  python tools/soak_streams.py <first seed> <count> [processes]  ->  a summary line; exit code 1 on any difference.  Every worker is a process of
its own (the glue chooses its backend once per process).
A stream whose output differs is decoded again: by the backend single-threaded (must then agree) and by the REFERENCE with the drawn thread count, six times — the
reference's own tile threads race on some streams (transform.cc:398 looks the prediction mode of a CHROMA block up at its chroma coordinates in the luma-indexed
array: with 4:2:2 and transform_skip_rotation that is a position in another tile column, which another thread may or may not have parsed yet; the glue's recorder
restates the lookup literally and inherits the race): where the reference with threads does not reproduce its own single-threaded output, the stream is counted as
"reference varies with threads", not as a difference."""
import ctypes
import multiprocessing as mp
import os
import random
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
STREAMGEN = os.path.join(ROOT, "oracle", "_ref", "streamgen")

F_WP, F_TSKIP, F_BYPASS, F_QPDELTA, F_PCM, F_SCALING, F_SCALING_PPS, F_REXT, F_CIP, F_DEPSLICE = 1, 2, 4, 8, 16, 32, 64, 256, 512, 1024
F_RA, F_WPP, F_TMVP, F_SDH, F_LT = 2048, 4096, 8192, 16384, 32768
F_MIXSLICE, F_LISTMOD, F_NOOUTPUT = 65536, 131072, 262144
G_CTB32, G_CTB16, G_MINCB16, G_TILES, G_NOTILEFILTER, G_PARMERGE, G_TB16, G_CONFWIN = 1, 2, 4, 8, 16, 32, 64, 128


def draw(seed):
    r = random.Random(77000 + seed)
    geom = 0
    ctb = r.choice([64, 64, 32, 16])
    geom |= {64: 0, 32: G_CTB32, 16: G_CTB16}[ctb]
    for bit, pr in ((G_MINCB16, .2), (G_NOTILEFILTER, .3), (G_PARMERGE, .3), (G_TB16, .2), (G_CONFWIN, .2)):
        if r.random() < pr and not (bit == G_MINCB16 and ctb == 16):
            geom |= bit
    big = int(os.environ.get("SOAK_BIG", "0"))            # SOAK_BIG=<n>: widths and heights n times as large (n = 4: up to 3328 x 2048)
    w = r.randrange(3, 14) * 64 * max(1, big)
    h = r.randrange(2, 9) * 64 * max(1, big)
    if geom & G_CONFWIN:
        h += 0
    bd = r.choice([8, 8, 10, 10, 9, 12])
    chroma = r.choice([1, 1, 1, 2, 3, 0])
    tc = r.randrange(1, min(3, w // ctb // 2 + 1) + 1) if r.random() < .5 else 1
    tr = r.randrange(1, min(2, h // ctb // 2 + 1) + 1) if r.random() < .5 else 1
    if tc * tr > 1 and r.random() < .4:
        geom |= G_TILES
    feat = 0
    ra = r.random() < .35
    for bit, pr in ((F_WP, .3), (F_TSKIP, .3), (F_BYPASS, .2), (F_QPDELTA, .4), (F_PCM, .25), (F_SCALING, .2), (F_SCALING_PPS, .15), (F_CIP, .25), (F_DEPSLICE, .25),
                    (F_TMVP, .5), (F_SDH, .4), (F_MIXSLICE, .2), (F_LISTMOD, .2), (F_NOOUTPUT, .15)):
        if r.random() < pr:
            feat |= bit
    if ra:
        feat |= F_RA
        if r.random() < .4 and tc * tr == 1:
            feat |= F_WPP
        if r.random() < .5:
            feat |= F_LT
    if chroma in (2, 3, 0) and bd != 8 or chroma in (2, 3):
        feat |= F_REXT
    frames = r.choice([9, 17]) if ra else r.randrange(2, 7)
    if big:
        frames = 9 if ra else min(frames, 3)
    slices = r.choice([1, 1, 2, 3])
    # the writer's own rules (oracle/ref_streamgen.cc main): the reference picture set and the collocated picture are per picture (one slice), WPP means one slice and
    # no tiles, the hidden sign is written outside the range extensions only
    if feat & (F_RA | F_TMVP | F_WPP):
        slices = 1
    if slices == 1:
        feat &= ~(F_MIXSLICE | F_DEPSLICE)
    if feat & F_REXT:
        feat &= ~F_SDH
    return dict(w=w, h=h, bd=bd, tc=tc, tr=tr, frames=frames, seed=seed, intra_pct=r.choice([3, 10, 30]), b_frames=r.choice([0, 1, 1]), sao=r.choice([0, 1, 1]),
                features=feat, chroma=chroma, slices=slices, geom=geom, threads=r.choice([0, 4, 8]))


def work(args):
    first, count, k, n = args
    import de265_py
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libde265_ref.so"))
    glue = ctypes.CDLL(os.path.join(ROOT, "glue", "_build", "libde265.so"))
    glue.m355_glue_cpu_pixel_calls.restype = ctypes.c_longlong
    done = refused = racy = 0
    bad = []
    tmp = tempfile.mkdtemp(prefix="soak_streams_")
    for seed in range(first + k, first + count, n):
        c = draw(seed)
        out = os.path.join(tmp, "s%d.h265" % seed)
        g = subprocess.run([STREAMGEN, out] + [str(c[x]) for x in ("w", "h", "bd", "tc", "tr", "frames", "seed", "intra_pct", "b_frames", "sao", "features", "chroma", "slices", "geom")],
                           capture_output=True)
        if g.returncode != 0 or not os.path.exists(out):
            refused += 1                                   # a combination the writer does not build
            continue
        data = open(out, "rb").read()
        os.unlink(out)
        if os.environ.get("SOAK_DAMAGE"):
            # SOAK_DAMAGE=1: the stream with bits flipped inside slice data / with pictures removed (tests/test_streams.py flip_bits, drop_pictures), decoded
            # single-threaded by both: the backend must show what the reference shows — the same samples, the same number of pictures, the same warnings
            import test_streams
            rr = random.Random(123457 + seed)
            data = test_streams.flip_bits(data, seed) if rr.random() < 0.6 else test_streams.drop_pictures(data, {rr.randrange(0, c["frames"]) for _ in range(rr.randrange(1, 3))})
            try:
                want = de265_py.decode_stream(ref, data, threads=0, scalar=True)
                got = de265_py.decode_stream(glue, data, threads=0)
                ok = got[:2] == want[:2] and set(got[2]) - {1000} == set(want[2]) - {1000} and glue.m355_glue_cpu_pixel_calls() == 0
            except Exception as e:                         # noqa: BLE001
                ok, want, got = False, ("exception",), ("exception", str(e)[:100])
            if not ok:
                bad.append((seed, c, want, got))
            done += 1
            continue
        if os.environ.get("SOAK_APP"):
            # SOAK_APP=1: the stream driven the ways applications drive a decoder (tests/test_glue_app_patterns.py): pushed in pieces of 1 byte .. 20 KB between decode
            # calls (pictures taken with get or with peek / release), NAL unit by NAL unit, or with a de265_reset somewhere in the middle followed by the stream from its
            # start — the reference single-threaded, the backend with the drawn thread count, the same calls in the same order
            import test_glue_app_patterns as ap
            rr = random.Random(424243 + seed)
            kind = rr.choice(["pieces", "pieces", "nal", "reset"])
            try:
                if kind == "pieces":
                    sd, peek = rr.randrange(1 << 20), rr.random() < 0.5
                    want, got = ap.chunked(ref, data, sd, peek, 0), ap.chunked(glue, data, sd, peek, c["threads"])
                elif kind == "nal":
                    want, got = ap.by_nal(ref, data, 0), ap.by_nal(glue, data, c["threads"])
                else:
                    cut = rr.randrange(1, len(data))
                    want, got = ap.with_reset(ref, data, cut, 0), ap.with_reset(glue, data, cut, c["threads"])
                ok = want[1] > 0 and got[:2] == want[:2] and glue.m355_glue_cpu_pixel_calls() == 0
            except Exception as e:                         # noqa: BLE001
                ok, want, got = False, ("exception",), (kind, str(e)[:100])
            if not ok:
                bad.append((seed, kind, c, want, got))
            done += 1
            continue
        try:
            want = de265_py.decode_stream(ref, data, threads=0, scalar=True)
        except Exception as e:                             # noqa: BLE001
            refused += 1
            continue
        if want[1] == 0 or want[2]:
            refused += 1                                   # the reference itself warns about / rejects the stream: not a case
            continue
        try:
            got = de265_py.decode_stream(glue, data, threads=c["threads"])
            ok = got[:2] == want[:2] and set(got[2]) <= {1000} and glue.m355_glue_cpu_pixel_calls() == 0
        except Exception as e:                             # noqa: BLE001
            ok, got = False, ("exception", str(e)[:100])
        if not ok and isinstance(got, tuple) and c["threads"] > 0 and glue.m355_glue_cpu_pixel_calls() == 0:
            try:
                single = de265_py.decode_stream(glue, data, threads=0)
                ref_thr = {de265_py.decode_stream(ref, data, threads=c["threads"])[:2] for _ in range(6)}
                if single[:2] == want[:2] and ref_thr != {want[:2]}:
                    ok = True; racy += 1
            except Exception as e:                         # noqa: BLE001
                pass
        if not ok:
            bad.append((seed, c, want[:2], got[:2] if isinstance(got, tuple) else got))
        done += 1
    return done, refused, bad, racy


if __name__ == "__main__":
    first, count = int(sys.argv[1]), int(sys.argv[2])
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    t0 = time.time()
    with mp.get_context("spawn").Pool(n) as pool:
        res = pool.map(work, [(first, count, k, n) for k in range(n)])
    done = sum(r[0] for r in res); refused = sum(r[1] for r in res); bad = [b for r in res for b in r[2]]; racy = sum(r[3] for r in res)
    print("soak_streams: seeds %d..%d, %d streams decoded by the reference and by the backend, %d not built / refused by the reference, %d where the reference itself varies with threads (backend single-threaded agrees), %d DIFFER, %.0f s on %d processes"
          % (first, first + count - 1, done, refused, racy, len(bad), time.time() - t0, n))
    for b in bad[:20]:
        print("  ", b)
    sys.exit(1 if bad else 0)
