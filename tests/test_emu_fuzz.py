"""The picture layer must be TOTAL on its input domain (slots/drivers have no error channel, bad input is the
caller's fault — but a GPU backend must never turn a corrupt list into wild memory traffic): randomly corrupted work lists
are either rejected by validation (M355_ERR_INVALID) or decode without a crash / hang / device timeout.  Runs the product
kernels under the SIMT interpreter, where an out-of-bounds access of a kernel would fault the test process."""
import numpy as np
import pytest

from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from synth_util import make_case
from libde265_amd import capi, worklist


def corrupt(pic, rng):
    """one random field of one random record set to a random (often out-of-range) value"""
    if rng.integers(8) == 0:
        # picture parameters: geometry / block sizes / bit depths / tile grid
        pp = pic.pp.copy()
        fields = [f for f in pp.dtype.names if not f.startswith("reserved")]
        f = fields[rng.integers(len(fields))]
        col = pp[f]
        info = np.iinfo(col.dtype)
        val = [info.max, info.min, int(rng.integers(info.min, int(info.max) + 1)), 0, 1, int(col.flat[0]) + 1, int(col.flat[0]) + 4][rng.integers(7)]
        val = min(max(val, info.min), info.max)
        col.flat[rng.integers(col.size)] = val
        pic.pp = pp
        return "pp.%s=%d" % (f, val)
    lists = [n for n in ("ctbs", "cus", "tus", "pbs", "rbs", "ibs", "slices", "wts") if len(getattr(pic, n))]
    name = lists[rng.integers(len(lists))]
    arr = getattr(pic, name).copy()
    fields = [f for f in arr.dtype.names if not f.startswith("reserved")]
    f = fields[rng.integers(len(fields))]
    i = rng.integers(len(arr))
    col = arr[f]
    info = np.iinfo(col.dtype)
    val = [info.max, info.min, rng.integers(info.min, info.max + 1), 0, 1][rng.integers(5)]
    if col.ndim == 1:
        col[i] = val
    else:
        col[i].flat[rng.integers(col[i].size)] = val
    setattr(pic, name, arr)
    return "%s[%d].%s=%d" % (name, i, f, val)


@pytest.mark.parametrize("in_place", [False, True, "resident"], ids=["copying_submit", "in_place_submit", "in_place_resident"])
@pytest.mark.parametrize("seed", range(8))
def test_corrupted_lists_never_crash(emu_lib, seed, in_place):  # noqa: F811
    """in_place: the lists are recorded into the pinned arena and their records are checked ON THE DEVICE (k_validate) — a rejected
    picture is never acted upon and the error surfaces at m355_wait.  "resident": recorded into a HANDLE's arena
    (m355_picture_arena_begin -> m355_picture_replace -> m355_decode_resident), the same handle for every picture."""
    rng = np.random.default_rng(1000 + seed)
    cfg = [dict(width=128, height=64, bit_depth=8, seed=301, tile_cols=2), dict(width=96, height=96, bit_depth=10, seed=302, intra_pct=60, features=31),
           dict(width=128, height=64, bit_depth=8, seed=303, chroma_format=3, features=32)][seed % 3]
    ctx = capi.Context(emu_lib, 0)
    try:
        accepted = rejected = 0
        handle = -1
        for _ in range(25):
            pic, refs = make_case(**cfg)
            pp = pic.pp[0].copy()
            what = [corrupt(pic, rng) for _ in range(1 + rng.integers(3))]
            handles = [ctx.frame_create_for(pp) for _ in refs]
            for h, planes in zip(handles, refs):
                ctx.frame_upload(h, planes)
            pic.dst_frame = ctx.frame_create_for(pp)
            pic.ref_frames = [handles[i] if i < len(handles) else -1 for i in range(worklist.MAX_REF_FRAMES)]
            try:
                if in_place == "resident":
                    handle = ctx.upload_in_place(pic, handle=handle, fill_threads=1)
                    ctx.decode_resident(handle)
                elif in_place:
                    ctx.submit_in_place(pic, fill_threads=1)
                else:
                    ctx.submit(pic)
                ctx.wait()
                accepted += 1
            except capi.M355Error as e:
                assert e.code in (3, 5), "unexpected failure %s for %s" % (e, what)     # INVALID (or a bounded device timeout)
                rejected += 1
            ctx.frame_destroy(pic.dst_frame)
            for h in handles:
                ctx.frame_destroy(h)
        assert accepted + rejected == 25
    finally:
        ctx.close()
