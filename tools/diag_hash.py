"""Times m355_frame_hash (CRC / checksum kernels, MD5 host path) on an 8K 10-bit frame.  Diagnostic only."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from libde265_amd import capi
from hash_util import make_planes

lib = capi.Library()
ctx = capi.Context(lib, 0)
for (w, h, bd) in ((7680, 4320, 10), (3840, 2160, 8), (1920, 1080, 8)):
    f = ctx.frame_create(w, h, 1, bd, bd)
    ctx.frame_upload(f, make_planes(w, h, 1, bd, bd, 1))
    nbytes = w * h * 3 // 2 * (2 if bd > 8 else 1)
    for t, name in ((1, "crc"), (2, "checksum"), (0, "md5(host)")):
        ctx.frame_hash(f, t)
        n = 20 if t else 2
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.frame_hash(f, t)
        dt = (time.perf_counter() - t0) / n
        print("%dx%d %2d-bit %-10s %8.3f ms per picture (call incl. sync + readback)  %7.1f GB/s" % (w, h, bd, name, dt * 1e3, nbytes / dt / 1e9))
    ctx.frame_destroy(f)
ctx.close()
