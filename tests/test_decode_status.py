"""Per-picture outcome of asynchronous submits (m355_decode_status / m355_last_serial).  Lists recorded in place are checked on the
DEVICE (k_validate) ahead of the picture's kernels, long after m355_submit_picture has returned: a rejected picture must be
attributable — its serial, the record — its kernels must not have acted on the lists (destination frame untouched), pictures before
and after it on the same lanes decode normally, and m355_wait reports every rejected picture once.

CPU tier: SIMT-interpreter build.  GPU tier: the product library — the gate rejects on real hardware."""
import numpy as np
import pytest

from oracle_py import Oracle
from synth_util import assert_planes_equal, make_case, oracle_decode
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from libde265_amd import capi, worklist

CASE = dict(width=256, height=192, bit_depth=8, seed=71, intra_pct=30, tile_cols=2, features=2)

# corruptions that only the record checks can see (the host's in-place checks cover the CTB table and PB geometry only)
CORRUPTIONS = [("cus", "pred_mode", 7, "cu"), ("tus", "log2_size", 9, "tu"), ("rbs", "coeff_ofs", 0x7FFFFFF0, "rb"), ("ibs", "mode", 77, "ib"),
               ("pbs", "ref_slot", 31, "pb"), ("rbs", "kind", 9, "rb")]


def run(lib, oracle, depth, ring=True):
    pic, refs = make_case(**CASE)
    pp = pic.pp[0]
    want = oracle_decode(Oracle(oracle), pic, refs)
    ctx = capi.Context(lib, 0)
    try:
        handles = []
        for planes in refs:
            f = ctx.frame_create_for(pp)
            ctx.frame_upload(f, planes)
            handles.append(f)
        pic.ref_frames = [handles[i] if i < len(handles) else -1 for i in range(worklist.MAX_REF_FRAMES)]
        ctx.set_pipeline_depth(depth)
        serials, dsts, bad_serials = [], [], {}
        for k in range(2 * len(CORRUPTIONS) + 1):
            d = ctx.frame_create_for(pp)
            ctx.frame_fill(d, 77, 99)
            dsts.append(d)
            p = pic
            if k % 2 == 1:
                lst, field, val, name = CORRUPTIONS[k // 2]
                p = make_case(**CASE)[0]
                p.ref_frames = pic.ref_frames
                arr = getattr(p, lst).copy()
                i = len(arr) // 2
                if arr[field].ndim == 1:
                    arr[field][i] = val
                else:
                    arr[field][i].flat[0] = val
                setattr(p, lst, arr)
                bad_serials[k + 1] = (name, i)
            p.dst_frame = d
            ctx.submit_in_place(p, fill_threads=1)
            serials.append(ctx.last_serial())
        assert serials == list(range(1, len(serials) + 1))
        for k, sn in enumerate(serials):
            st = ctx.decode_status(sn)
            while st == 6:                                  # M355_ERR_BUSY
                st = ctx.decode_status(sn)
            planes = ctx.frame_download(dsts[k])
            if sn in bad_serials:
                assert st == 3, "picture %d (corrupted %s) was not rejected" % (sn, bad_serials[sn])
                msg = lib.error()
                assert "picture %d:" % sn in msg and bad_serials[sn][0] + " " in msg, msg
                assert all((pl == (77 if c == 0 else 99)).all() for c, pl in enumerate(planes)), "a rejected picture's kernels wrote its destination frame"
            else:
                assert st == 0, lib.error()
                assert_planes_equal(planes, want, "picture %d beside rejected ones" % sn)
        # m355_wait: everything was reported through m355_decode_status already
        ctx.wait()
        # ... and reports what nobody asked about
        bad = make_case(**CASE)[0]
        bad.ref_frames = pic.ref_frames
        arr = bad.ibs.copy(); arr["mode"][0] = 99; bad.ibs = arr
        bad.dst_frame = dsts[0]
        ctx.submit_in_place(bad, fill_threads=1)
        with pytest.raises(capi.M355Error) as e:
            ctx.wait()
        assert e.value.code == 3 and "picture %d:" % (serials[-1] + 1) in str(e.value)
        ctx.wait()                                          # reported once
        # a rejection nobody asks about must survive more submits than the status ring holds (64); the fillers are tiny pictures
        if ring:
            bad.dst_frame = dsts[1]
            ctx.submit_in_place(bad, fill_threads=1)
            lost = ctx.last_serial()
            small, srefs = make_case(width=64, height=64, bit_depth=8, seed=72, intra_pct=30)
            spp = small.pp[0]
            sh = []
            for planes in srefs:
                f = ctx.frame_create_for(spp)
                ctx.frame_upload(f, planes)
                sh.append(f)
            small.ref_frames = [sh[i] if i < len(sh) else -1 for i in range(worklist.MAX_REF_FRAMES)]
            small.dst_frame = ctx.frame_create_for(spp)
            for _ in range(70):
                ctx.submit_in_place(small, fill_threads=1)
            assert ctx.decode_status(lost) == 7 and "no longer kept" in lib.error()      # M355_ERR_STALE, not a rejection
            with pytest.raises(capi.M355Error) as e:
                ctx.wait()
            assert e.value.code == 3 and "picture %d" % lost in str(e.value) and "ring" in str(e.value)
            ctx.wait()
    finally:
        ctx.close()


@pytest.mark.parametrize("depth", [1, 3])
def test_decode_status_emulated(emu_lib, oracle, depth):  # noqa: F811
    run(emu_lib, oracle, depth, ring=depth == 3)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [1, 2, 4])
def test_decode_status_gpu(oracle, depth):
    run(capi.Library(), oracle, depth)
