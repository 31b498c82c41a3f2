#!/usr/bin/env python3
"""Generates tests/golden/girlshy*.m355gold.gz from the REAL reference (run in the build container,
where /root/reference exists):

    python tests/golden/make_girlshy_fixture.py

It runs the reference decoder on testdata/girlshy.h265 through oracle/ref_recorder.cc (built into
oracle/_ref/libde265_ref.so by `make -C oracle ref`), which records the per-picture work lists the
pixel path consumes plus the reference's own final planes, and stores
  * the work lists (decode order),
  * the MD5 of each plane of each picture as the reference produced it,
  * the display order + conformance crop, so that the whole-stream YUV MD5 can be re-derived and
    compared with the reference CI's golden b81538fa33a67278e5263e231e43ca98 (scripts/ci-run.sh:91-92).
Four variants: the full chain, SAO disabled, deblocking disabled, and deblocking+SAO disabled (dec265 --disable-deblocking --disable-sao,
golden 098a8f4d62bef69504174073879cd4ad measured with the reference in SURVEY.md 8c).
"""
import ctypes
import gzip
import hashlib
import json
import os
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from libde265_amd import worklist  # noqa: E402

STREAM = "/root/reference/testdata/girlshy.h265"
GOLDEN = {"full": "b81538fa33a67278e5263e231e43ca98", "nolf": "098a8f4d62bef69504174073879cd4ad",
          # stage-isolated goldens measured with the reference CLI: dec265 --disable-sao / --disable-deblocking
          "nosao": "f0647c472db1a58c4a5f8b605da7513b", "nodeblk": "6983e435cf17979b90b721555a45493b"}


def record_fixture(ref, stream, out, nodeblk, nosao, meta, expect_md5=None):
    """run the reference decoder on `stream` through ref_recorder and write the fixture `out`; returns the stream MD5"""
    with tempfile.TemporaryDirectory() as td:
        rec = os.path.join(td, "rec")
        n = ref.ref_record_stream(stream.encode(), rec.encode(), nodeblk, nosao)
        assert n > 0, n
        raw = open(rec, "rb").read()
        planes = open(rec + ".planes", "rb").read()
        order = [[int(v) for v in l.split()] for l in open(rec + ".order")]
    assert raw[:8] == b"M355REC1"
    npic = struct.unpack_from("<i", raw, 8)[0]
    o, po = 12, 0
    pics, blobs = [], []
    by_poc = {}
    for i in range(npic):
        poc, dpb, _, _ = struct.unpack_from("<4i", raw, o); o += 16
        pic, o2 = worklist.Picture.loads(raw, o)
        blobs.append(raw[o:o2]); o = o2
        pp = pic.pp[0]
        dims = worklist.plane_dims(int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"]))
        md5s, pl = [], []
        for c, (w, h) in enumerate(dims):
            bpp = 1 if (pp["bit_depth_luma"] if c == 0 else pp["bit_depth_chroma"]) <= 8 else 2
            nb = w * h * bpp
            md5s.append(hashlib.md5(planes[po:po + nb]).hexdigest()); pl.append((planes[po:po + nb], w, h, bpp)); po += nb
        pics.append({"poc": poc, "dpb": dpb, "md5": md5s})
        by_poc.setdefault(poc, []).append(pl)   # IDR-only streams repeat POC 0: first decoded is first output
    assert po == len(planes)
    # re-derive the whole-stream MD5 exactly as `dec265 -o -` writes it (cropped planes, display order)
    m = hashlib.md5()
    for poc, w, h, cx, cy in order:
        cur = by_poc[poc].pop(0)
        for c, (data, pw, ph, bpp) in enumerate(cur):
            sx = 1 if c == 0 else cur[0][1] // pw
            sy = 1 if c == 0 else cur[0][2] // ph
            cw, ch, ox, oy = w // sx, h // sy, cx // sx, cy // sy
            for y in range(ch):
                s = ((oy + y) * pw + ox) * bpp
                m.update(data[s:s + cw * bpp])
    stream_md5 = m.hexdigest()
    if expect_md5 is not None:
        assert stream_md5 == expect_md5, (out, stream_md5)
    hdr = dict(meta)
    hdr.update({"stream_md5": stream_md5, "pictures": pics, "order": order})
    hdr = json.dumps(hdr).encode()
    with gzip.GzipFile(out, "wb", compresslevel=9, mtime=0) as f:
        f.write(b"M355GOLD" + struct.pack("<I", len(hdr)) + hdr + b"".join(blobs))
    print(os.path.basename(out), npic, "pictures, stream md5", stream_md5, os.path.getsize(out), "bytes")
    return stream_md5


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-j8", "ref"], check=True)
    ref = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libde265_ref.so"))
    for variant, (nodeblk, nosao) in {"full": (0, 0), "nolf": (1, 1), "nosao": (0, 1), "nodeblk": (1, 0)}.items():
        record_fixture(ref, STREAM, os.path.join(HERE, "girlshy_%s.m355gold.gz" % variant), nodeblk, nosao,
                       {"stream": "testdata/girlshy.h265", "variant": variant}, GOLDEN[variant])


if __name__ == "__main__":
    main()
