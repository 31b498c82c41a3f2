"""ctypes front-end of oracle/ref_replay.cc (m355_ref_replay in oracle/_ref/libde265_ref.so): one picture's work
lists through the REAL reference functions.  Test infrastructure only."""
import ctypes

import numpy as np

from libde265_amd import worklist


def ref_replay(ref_lib, pic, refs, stages=worklist.STAGE_ALL, accel=0):
    """refs: list of plane lists (slot i = refs[i]); returns the replayed planes (tight numpy arrays)."""
    pp = pic.pp[0]
    w, h, cf = int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"])
    dims = [d for d in worklist.plane_dims(w, h, cf) if d[0]]
    dt = np.uint8 if int(pp["bit_depth_luma"]) <= 8 else np.uint16
    rp = (ctypes.c_void_p * (worklist.MAX_REF_FRAMES * 3))()
    keep = []
    for s, planes in enumerate(refs):
        for c, a in enumerate(planes):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            rp[s * 3 + c] = a.ctypes.data
    out = [np.zeros((ph, pw), dt) for (pw, ph) in dims]
    op = (ctypes.c_void_p * 3)(*([o.ctypes.data for o in out] + [None] * (3 - len(out))))
    saved = pic.ref_frames
    pic.ref_frames = [i if i < len(refs) else -1 for i in range(worklist.MAX_REF_FRAMES)]
    cpic, k2 = pic.to_c()
    pic.ref_frames = saved
    ref_lib.m355_ref_replay.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    ref_lib.m355_ref_replay.restype = ctypes.c_int
    rc = ref_lib.m355_ref_replay(ctypes.byref(cpic), rp, stages, accel, op)
    del keep, k2
    if rc != 0:
        raise RuntimeError("m355_ref_replay failed: %d" % rc)
    return out
