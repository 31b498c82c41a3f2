#!/usr/bin/env python3
"""Generates tests/golden/encintra_*.m355gold.gz: more REAL HEVC bitstreams, produced by the reference's own
(experimental) encoder and decoded by the reference decoder through oracle/ref_recorder.cc (run in the build
container, where /root/reference exists):

    python tests/golden/make_enc_fixture.py

The reference encoder only works for intra pictures at this commit (its inter analysis has a use-after-free,
encoder/algo/tb-intrapredmode.cc:509 reached from cb-intra-inter.cc:110; its CLI crashes before that, see
oracle/ref_encode.cc), so these streams pin the intra + residual chain on real encoder decisions: all 35 prediction
modes incl. mode-dependent scan/DST and implicit filters, NxN partitions, residual quadtrees down to 4x4 with
chroma 4x4 deferral, CTB sizes 16/32/64, pictures that end in partial CTBs.  8-bit 4:2:0, no deblocking / SAO
(encoder-context.cc:162-165).  Each fixture stores the work lists + the reference's plane MD5s, and the stream MD5 is
cross-checked against what the reference CLI (oracle/_ref/dec265 -o) writes for the same stream.
"""
import ctypes
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from make_girlshy_fixture import record_fixture  # noqa: E402

# name: (width, height, frames, seed, encoder options)
STREAMS = {
    "ctb64": (416, 240, 2, 11, ["--max-cb-size", "64", "--min-cb-size", "8", "-q", "26"]),
    "ctb32_hq": (176, 144, 2, 12, ["--max-cb-size", "32", "--min-cb-size", "8", "-q", "10", "--TB-IntraPredMode", "brute-force"]),
    "ctb16_lq": (200, 120, 3, 13, ["--max-cb-size", "16", "--min-cb-size", "8", "--max-tb-size", "16", "-q", "38"]),
}


def synth_yuv(path, w, h, n, seed):
    """moving textured content with flat areas, gradients and hard edges (so the encoder uses many modes / TU depths)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    tex = rng.integers(0, 256, (h + 64, w + 64)).astype(np.float64)
    for _ in range(3):   # cheap blur
        tex = (tex + np.roll(tex, 1, 0) + np.roll(tex, 1, 1) + np.roll(tex, -1, 0) + np.roll(tex, -1, 1)) / 5.0
    with open(path, "wb") as f:
        for k in range(n):
            t = tex[2 * k:2 * k + h, 3 * k:3 * k + w]
            y = 110 + 50 * np.sin((xx + 5 * k) / 19.0) * np.cos(yy / 13.0) + 2.5 * (t - 128)
            y += ((((xx + 3 * k) // 24).astype(int) + (yy // 40).astype(int)) % 2) * 35
            y += np.where((xx - w / 2 - 4 * k) * 0.7 + (yy - h / 2) > 0, 25, -10)        # a diagonal edge
            y[: h // 5, : w // 3] = 60                                                   # a flat area
            y = np.clip(y, 0, 255).astype(np.uint8)
            cx, cy = xx[::2, ::2], yy[::2, ::2]
            u = np.clip(128 + 45 * np.sin((cx + 4 * k) / 23.0) + 0.8 * (t[::2, ::2] - 128), 0, 255).astype(np.uint8)
            v = np.clip(128 + 45 * np.cos((cy + 2 * k) / 17.0) + (cx > w / 3) * 20, 0, 255).astype(np.uint8)
            f.write(y.tobytes()); f.write(u.tobytes()); f.write(v.tobytes())


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-j8", "ref", "enc"], check=True)
    refdir = os.path.join(ROOT, "oracle", "_ref")
    ref = ctypes.CDLL(os.path.join(refdir, "libde265_ref.so"))
    for name, (w, h, n, seed, opts) in STREAMS.items():
        with tempfile.TemporaryDirectory() as td:
            yuv, bits, dec = (os.path.join(td, x) for x in ("in.yuv", "out.bin", "dec.yuv"))
            synth_yuv(yuv, w, h, n, seed)
            subprocess.run([os.path.join(refdir, "ref_encode"), yuv, str(w), str(h), str(n), bits, "--sop-structure", "intra"] + opts,
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            subprocess.run([os.path.join(refdir, "dec265"), "-q", "-o", dec, bits], check=True, stdout=subprocess.DEVNULL)
            cli_md5 = hashlib.md5(open(dec, "rb").read()).hexdigest()
            src = np.fromfile(yuv, np.uint8).astype(np.float64)
            out = np.fromfile(dec, np.uint8).astype(np.float64)
            assert src.size == out.size, (src.size, out.size)
            psnr = 10 * np.log10(255.0 ** 2 / max(np.mean((src - out) ** 2), 1e-9))
            record_fixture(ref, bits, os.path.join(HERE, "encintra_%s.m355gold.gz" % name), 0, 0,
                           {"stream": "reference encoder, intra, %dx%d x%d, %s" % (w, h, n, " ".join(opts)),
                            "variant": name, "bitstream_bytes": os.path.getsize(bits), "psnr": round(psnr, 2)},
                           expect_md5=cli_md5)
            print("   ", name, "bitstream", os.path.getsize(bits), "bytes, PSNR %.2f dB" % psnr)


if __name__ == "__main__":
    main()
