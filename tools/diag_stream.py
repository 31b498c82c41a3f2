#!/usr/bin/env python3
"""Where does a generated stream (tools/soak_streams.py seed) come out different?  For each seed: the reference's pictures against the backend's, decoded <reps>
times with the drawn thread count and once single-threaded; for the first picture that differs: plane, number of samples, bounding box, a few values.
python tools/diag_stream.py <reps> <seed> [seed ...]   (environment: M355_PIPELINE_DEPTH, M355_GLUE_SYNC ... as the glue reads them)"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import de265_py          # noqa: E402
import soak_streams as S  # noqa: E402


def first_diff(want, got):
    for k, (a, b) in enumerate(zip(want, got)):
        for c, (x, y) in enumerate(zip(a, b)):
            if x.shape != y.shape:
                return "picture %d plane %d: shape %s against %s" % (k, c, x.shape, y.shape)
            d = np.argwhere(x != y)
            if len(d):
                y0, x0 = d.min(0); y1, x1 = d.max(0)
                ex = ["(%d,%d): %d->%d" % (q[1], q[0], x[q[0], q[1]], y[q[0], q[1]]) for q in d[:6]]
                return "picture %d plane %d: %d samples differ, x %d..%d y %d..%d  %s" % (k, c, len(d), x0, x1, y0, y1, " ".join(ex))
    if len(want) != len(got):
        return "%d pictures against %d" % (len(want), len(got))
    return None


if __name__ == "__main__":
    reps = int(sys.argv[1])
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libde265_ref.so"))
    glue = ctypes.CDLL(os.path.join(ROOT, "glue", "_build", "libde265.so"))
    for seed in map(int, sys.argv[2:]):
        c = S.draw(seed)
        out = "/tmp/diag_s%d.h265" % seed
        subprocess.run([S.STREAMGEN, out] + [str(c[x]) for x in ("w", "h", "bd", "tc", "tr", "frames", "seed", "intra_pct", "b_frames", "sao", "features", "chroma", "slices", "geom")], capture_output=True)
        data = open(out, "rb").read()
        if os.environ.get("SOAK_DAMAGE"):                   # (the damage tools/soak_streams.py applies to this seed)
            import random
            import test_streams
            rr = random.Random(123457 + seed)
            data = test_streams.flip_bits(data, seed) if rr.random() < 0.6 else test_streams.drop_pictures(data, {rr.randrange(0, c["frames"]) for _ in range(rr.randrange(1, 3))})
        want = []
        de265_py.decode_stream(ref, data, threads=0, scalar=True, planes_out=want)
        print("seed %d %s" % (seed, c))
        for th in ([c["threads"]] * reps if not os.environ.get("SOAK_DAMAGE") else []) + [0]:
            got = []
            try:
                de265_py.decode_stream(glue, data, threads=th, planes_out=got)
                d = first_diff(want, got)
            except Exception as e:      # noqa: BLE001
                d = "exception %s" % str(e)[:100]
            print("   threads %d: %s" % (th, d or "identical"))
