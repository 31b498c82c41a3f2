"""Shared driver for synthetic-picture parity: same work lists + same reference planes through the
oracle and through a libde265_mi355x build (real HIP library or the SIMT-interpreter build)."""
import numpy as np

from libde265_amd import synth, worklist


def make_case(**cfgkw):
    pic = synth.picture(**cfgkw)
    pp = pic.pp[0]
    n_refs = pic.meta["cfg"]["n_refs"]
    refs = [synth.ref_planes(pic.meta["cfg"]["seed"] + 17 * i, int(pp["width"]), int(pp["height"]),
                             int(pp["chroma_format_idc"]), int(pp["bit_depth_luma"])) for i in range(n_refs)]
    return pic, refs


def oracle_decode(o, pic, refs, stages=worklist.STAGE_ALL):
    pp = pic.pp[0]
    rframes = {}
    for i, planes in enumerate(refs):
        f = o.frame_new(pp)
        o.frame_set_planes(f, planes)
        rframes[i] = f
    dst = o.frame_new(pp)
    pic.ref_frames = [i if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
    assert o.decode(pic, dst, rframes, stages) == 0
    out = o.frame_planes(dst)
    o.frame_free(dst)
    for f in rframes.values():
        o.frame_free(f)
    return out


def device_decode(ctx, pic, refs, stages=worklist.STAGE_ALL, resident=False, repeat=1):
    pp = pic.pp[0]
    handles = []
    for planes in refs:
        f = ctx.frame_create_for(pp)
        ctx.frame_upload(f, planes)
        handles.append(f)
    dst = ctx.frame_create_for(pp)
    pic.dst_frame = dst
    pic.ref_frames = [handles[i] if i < len(handles) else -1 for i in range(worklist.MAX_REF_FRAMES)]
    ctx.set_stages(stages)
    if resident:
        h = ctx.upload(pic)
        for _ in range(repeat):
            ctx.decode_resident(h)
        ctx.wait()
        ctx.release(h)
    else:
        for _ in range(repeat):
            ctx.submit(pic)
        ctx.wait()
    out = ctx.frame_download(dst)
    ctx.frame_destroy(dst)
    for f in handles:
        ctx.frame_destroy(f)
    ctx.set_stages(worklist.STAGE_ALL)
    return out


def assert_planes_equal(got, want, what=""):
    assert len(got) == len(want)
    for c, (g, w) in enumerate(zip(got, want)):
        if not np.array_equal(g, w):
            d = np.argwhere(g != w)
            raise AssertionError("%s plane %d: %d samples differ, first at (y,x)=%s, got %d want %d, bbox %s..%s" %
                                 (what, c, len(d), tuple(d[0]), g[tuple(d[0])], w[tuple(d[0])], d.min(0), d.max(0)))
