#!/usr/bin/env python3
"""Several decoders of ONE process at the same time, each on a thread of its own (the pattern of a transcoding server; ctypes releases the interpreter lock inside the
decoder's calls): every thread decodes its own generated stream (tools/soak_streams.py draw) <reps> times through glue/_build/libde265.so on the product backend — with
the parser threads its draw asks for — and compares with the reference's single-threaded decode.  python tools/soak_concurrent.py <reps> <seed> [seed ...]"""
import sys, os, ctypes, subprocess, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_streams as S, de265_py
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libde265_ref.so"))
glue = ctypes.CDLL(os.path.join(ROOT, "glue", "_build", "libde265.so"))
de265_py.bind(ref); de265_py.bind(glue)
seeds = list(map(int, sys.argv[2:]))
streams = {}
for seed in seeds:
    c = S.draw(seed)
    out = "/tmp/cs%d.h265" % seed
    subprocess.run([S.STREAMGEN, out] + [str(c[x]) for x in ("w", "h", "bd", "tc", "tr", "frames", "seed", "intra_pct", "b_frames", "sao", "features", "chroma", "slices", "geom")], capture_output=True)
    data = open(out, "rb").read()
    streams[seed] = (c, data, de265_py.decode_stream(ref, data, threads=0, scalar=True)[:2])
bad = []
def run(seed, reps):
    c, data, want = streams[seed]
    for i in range(reps):
        try:
            got = de265_py.decode_stream(glue, data, threads=c["threads"])[:2]
        except Exception as e:
            got = ("exception", str(e)[:80])
        if got != want:
            bad.append((seed, i, want, got))
ths = [threading.Thread(target=run, args=(s, int(sys.argv[1]))) for s in seeds]
[t.start() for t in ths]; [t.join() for t in ths]
print("concurrent decoders in one process: %d threads x %s decodes, %d differ" % (len(seeds), sys.argv[1], len(bad)), bad[:5])
sys.exit(1 if bad else 0)
