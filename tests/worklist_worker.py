"""Parity cases with k_intra's work list made on the DEVICE (M355_DEVICE_WORKLIST, read once per process): python worklist_worker.py
<library .so or "default"> <oracle .so>.  Mode 2 makes the library compare every device-made list with the host's (upload() fails on a
difference); mode 1 decodes from the device-made list alone.  Exit code 0 = every picture equals the oracle's."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from libde265_amd import capi, worklist  # noqa: E402
from oracle_py import Oracle  # noqa: E402
from shard_util import group_sharded_decode  # noqa: E402
from synth_util import assert_planes_equal, make_case, oracle_decode  # noqa: E402

CASES = [dict(width=256, height=192, bit_depth=8, seed=901, tile_cols=2, tile_rows=2, intra_pct=40),          # dependent CTBs across the wavefronts of four tiles
         dict(width=192, height=128, bit_depth=10, seed=902, intra_pct=100, n_refs=0),                          # an intra picture: every CTB has blocks
         dict(width=320, height=128, bit_depth=8, seed=903, n_slices=3, tile_cols=3, intra_pct=15, features=31),   # slices x tiles: the 3x3 neighbourhood facts
         dict(width=200, height=136, bit_depth=8, seed=904, log2_ctb=4, intra_pct=25),                          # 16x16 CTBs, ragged right / bottom edge
         dict(width=128, height=64, bit_depth=8, seed=905, intra_pct=0)]                                          # no intra block at all
BIG = [dict(width=1920, height=1080, bit_depth=8, seed=906, tile_cols=2, tile_rows=2, intra_pct=10), dict(width=3840, height=2160, bit_depth=10, seed=907, tile_cols=4, tile_rows=2, intra_pct=5)]


def one(lib, o, case):
    pic, refs = make_case(**case)
    pp = pic.pp[0]
    want = oracle_decode(o, pic, refs)
    ctx = capi.Context(lib, 0)
    try:
        handles = []
        for planes in refs:
            f = ctx.frame_create_for(pp); ctx.frame_upload(f, planes); handles.append(f)
        pic.ref_frames = [handles[i] if i < len(handles) else -1 for i in range(worklist.MAX_REF_FRAMES)]
        ctx.set_pipeline_depth(2)
        dsts = [ctx.frame_create_for(pp) for _ in range(3)]
        pic.dst_frame = dsts[0]; ctx.submit(pic)                                   # copying submit
        pic.dst_frame = dsts[1]; ctx.submit_in_place(pic, slack=1.3, fill_threads=2)   # lists recorded in place
        pic.dst_frame = dsts[2]; h = ctx.upload(pic); ctx.decode_resident(h)       # resident
        ctx.wait()
        for d in dsts:
            assert_planes_equal(ctx.frame_download(d), want, "seed %d" % case["seed"])
    finally:
        ctx.close()


if __name__ == "__main__":
    big = sys.argv[1] == "default"
    lib = capi.Library() if big else capi.Library(sys.argv[1])
    o = Oracle(ctypes.CDLL(sys.argv[2]))
    for case in CASES + (BIG if big else []):
        one(lib, o, case)
    # tile-sharded ranks make their own lists (CTBs of other ranks hold no blocks)
    pic, refs = make_case(width=256, height=192, bit_depth=8, seed=908, tile_cols=2, tile_rows=2, intra_pct=30)
    want = oracle_decode(o, pic, refs)
    for r, got in enumerate(group_sharded_decode(lib, pic, refs, 4, in_place=True)):
        assert_planes_equal(got, want, "group rank %d" % r)
    print("work-list worker ok (M355_DEVICE_WORKLIST=%s)" % os.environ.get("M355_DEVICE_WORKLIST", "unset"))
