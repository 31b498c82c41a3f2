"""ctypes front-end of the CPU oracle (oracle/liboracle.so) — test infrastructure only."""
import ctypes
import hashlib

import numpy as np

from libde265_amd import worklist


class OFrame(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("chroma_format", ctypes.c_int),
                ("bd_luma", ctypes.c_int), ("bd_chroma", ctypes.c_int),
                ("w", ctypes.c_int * 3), ("h", ctypes.c_int * 3), ("stride", ctypes.c_ssize_t * 3),
                ("p", ctypes.c_void_p * 3)]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.o_frame_new.restype = ctypes.POINTER(OFrame)
        lib.o_frame_new.argtypes = [ctypes.c_int] * 5
        lib.o_frame_free.argtypes = [ctypes.POINTER(OFrame)]
        lib.o_frame_import.argtypes = [ctypes.POINTER(OFrame), ctypes.c_int, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int]
        lib.o_frame_export.argtypes = [ctypes.POINTER(OFrame), ctypes.c_int, ctypes.c_void_p, ctypes.c_ssize_t, ctypes.c_int]
        lib.o_decode_picture.argtypes = [ctypes.c_void_p, ctypes.POINTER(OFrame), ctypes.c_void_p, ctypes.c_int]

    def frame_new(self, pp):
        return self.lib.o_frame_new(int(pp["width"]), int(pp["height"]), int(pp["chroma_format_idc"]),
                                    int(pp["bit_depth_luma"]), int(pp["bit_depth_chroma"]))

    def frame_free(self, f):
        self.lib.o_frame_free(f)

    def frame_planes(self, f):
        """list of numpy planes (uint8 for 8-bit, uint16 otherwise), tight (w x h)."""
        out = []
        for c in range(3):
            w, h = f.contents.w[c], f.contents.h[c]
            if w == 0:
                continue
            bd = f.contents.bd_luma if c == 0 else f.contents.bd_chroma
            a = np.zeros((h, w), np.uint8 if bd <= 8 else np.uint16)
            self.lib.o_frame_export(f, c, a.ctypes.data, w, a.itemsize)
            out.append(a)
        return out

    def frame_set_planes(self, f, planes):
        for c, a in enumerate(planes):
            a = np.ascontiguousarray(a)
            self.lib.o_frame_import(f, c, a.ctypes.data, a.shape[1], a.itemsize)

    def decode(self, pic, dst, refs, stages=worklist.STAGE_ALL):
        """refs: dict slot -> frame pointer"""
        cpic, keep = pic.to_c()
        arr = (ctypes.POINTER(OFrame) * worklist.MAX_REF_FRAMES)()
        for slot, fr in refs.items():
            arr[slot] = fr
        rc = self.lib.o_decode_picture(ctypes.byref(cpic), dst, ctypes.cast(arr, ctypes.c_void_p), stages)
        del keep
        return rc


def plane_md5s(planes):
    return [hashlib.md5(np.ascontiguousarray(p).tobytes()).hexdigest() for p in planes]
