"""Loader for tests/golden/*.m355gold.gz (written by tests/golden/make_girlshy_fixture.py)."""
import gzip
import json
import os
import struct

from libde265_amd import worklist

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_gold(name):
    raw = gzip.open(os.path.join(GOLDEN_DIR, name), "rb").read()
    assert raw[:8] == b"M355GOLD"
    n = struct.unpack_from("<I", raw, 8)[0]
    hdr = json.loads(raw[12:12 + n].decode())
    o = 12 + n
    pics = []
    for meta in hdr["pictures"]:
        pic, o = worklist.Picture.loads(raw, o)
        pic.meta = meta
        pics.append(pic)
    return hdr, pics
