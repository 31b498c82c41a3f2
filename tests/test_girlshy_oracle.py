"""Picture-level pin of the oracle: the work lists recorded from the REAL reference decoder on
testdata/girlshy.h265 (75 pictures I/P/B, WPP, SAO, weighted prediction, CTB64) replayed through
o_decode_picture() must reproduce every plane of every picture the reference produced, and therefore
the reference CI's golden MD5 b81538fa33a67278e5263e231e43ca98 (scripts/ci-run.sh:91-92)."""
import hashlib

import pytest

from golden_io import load_gold
from oracle_py import Oracle, plane_md5s
from libde265_amd import worklist


def replay(decode_fn, new_frame, free_frame, get_planes, pics):
    """decode every picture in decode order, managing frames by DPB index like the reference's DPB"""
    frames = {}
    for i, pic in enumerate(pics):
        dst_idx = pic.dst_frame
        if dst_idx in frames:
            free_frame(frames.pop(dst_idx))
        dst = new_frame(pic.pp[0])
        refs = {s: frames[s] for s in range(worklist.MAX_REF_FRAMES) if pic.ref_frames[s] >= 0}
        assert decode_fn(pic, dst, refs) == 0
        frames[dst_idx] = dst
        yield i, pic, get_planes(dst)
    for f in frames.values():
        free_frame(f)


GOLDEN = {"full": "b81538fa33a67278e5263e231e43ca98", "nolf": "098a8f4d62bef69504174073879cd4ad",
          # stage-isolated (dec265 --disable-sao / --disable-deblocking)
          "nosao": "f0647c472db1a58c4a5f8b605da7513b", "nodeblk": "6983e435cf17979b90b721555a45493b"}


@pytest.mark.parametrize("variant", sorted(GOLDEN))
def test_girlshy_oracle_matches_reference(oracle, variant):
    hdr, pics = load_gold("girlshy_%s.m355gold.gz" % variant)
    assert hdr["stream_md5"] == GOLDEN[variant]
    o = Oracle(oracle)
    by_poc = {}
    for i, pic, planes in replay(o.decode, o.frame_new, o.frame_free, o.frame_planes, pics):
        assert plane_md5s(planes) == pic.meta["md5"], "picture %d (POC %d) differs from the reference" % (i, pic.meta["poc"])
        by_poc[pic.meta["poc"]] = planes
    # whole-stream MD5 exactly as `dec265 -o -` writes it: cropped planes in display order
    m = hashlib.md5()
    for poc, w, h, cx, cy in hdr["order"]:
        pl = by_poc[poc]
        for c, p in enumerate(pl):
            sx = 1 if c == 0 else pl[0].shape[1] // p.shape[1]
            sy = 1 if c == 0 else pl[0].shape[0] // p.shape[0]
            m.update(p[cy // sy:cy // sy + h // sy, cx // sx:cx // sx + w // sx].tobytes())
    assert m.hexdigest() == hdr["stream_md5"]
