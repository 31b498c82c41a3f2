"""Shared helpers of the picture-hash tests: oracle / reference bindings and the plane generator."""
import ctypes

import numpy as np

MD5, CRC, CHECKSUM = 0, 1, 2


def oracle_hash(olib, plane, bit_depth, hash_type):
    a = np.ascontiguousarray(plane)
    h, w = a.shape
    vp = a.ctypes.data_as(ctypes.c_void_p)
    olib.o_hash_checksum.restype = ctypes.c_uint32
    olib.o_hash_crc.restype = ctypes.c_uint32
    if hash_type == MD5:
        out = (ctypes.c_uint8 * 16)()
        olib.o_hash_md5(vp, w, h, ctypes.c_ssize_t(w), bit_depth, out)
        return bytes(out)
    f = olib.o_hash_crc if hash_type == CRC else olib.o_hash_checksum
    return int(f(vp, w, h, ctypes.c_ssize_t(w), bit_depth))


def make_planes(w, h, cf, bdl, bdc, seed):
    """random planes with structure (so that position-dependent terms matter): noise + gradient, full value range"""
    rng = np.random.default_rng(seed)
    sw, sh = (2 if cf in (1, 2) else 1), (2 if cf == 1 else 1)
    out = []
    for c in range(3 if cf else 1):
        pw, ph, bd = (w, h, bdl) if c == 0 else (w // sw, h // sh, bdc)
        a = rng.integers(0, 1 << bd, (ph, pw), dtype=np.int64)
        a[: ph // 3] = (a[: ph // 3] + np.arange(pw)[None, :] * 3) % (1 << bd)
        out.append(a.astype(np.uint8 if bd <= 8 else np.uint16))
    return out


def ref_check(ref, w, h, cf, bdl, bdc, planes, hash_type, values):
    """the reference's own verdict (process_sei) on candidate hash values: 0 = accepted"""
    arr = (ctypes.c_void_p * 3)(*[p.ctypes.data for p in planes] + [None] * (3 - len(planes)))
    md5 = (ctypes.c_uint8 * 48)()
    crc = (ctypes.c_uint16 * 3)()
    chk = (ctypes.c_uint32 * 3)()
    for c, v in enumerate(values):
        if hash_type == MD5:
            md5[16 * c:16 * c + 16] = list(v)
        elif hash_type == CRC:
            crc[c] = v
        else:
            chk[c] = v
    return ref.m355_ref_check_hash(w, h, cf, bdl, bdc, arr, hash_type, md5, crc, chk)
