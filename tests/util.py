"""Shared test helpers: the xorshift32 generator the reference's dev-tools tests use
(dev-tools/test-transform.cc et al.: s^=s<<13; s^=s>>17; s^=s<<5) and ctypes glue."""
import ctypes

import numpy as np


class XorShift32:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFF
        assert self.s != 0

    def next(self):
        s = self.s
        s ^= (s << 13) & 0xFFFFFFFF
        s ^= s >> 17
        s ^= (s << 5) & 0xFFFFFFFF
        self.s = s
        return s

    def below(self, n):
        return self.next() % n

    def range(self, lo, hi):
        return lo + self.next() % (hi - lo + 1)

    def array(self, n, lo, hi, dtype):
        return np.array([self.range(lo, hi) for _ in range(n)], dtype=dtype)


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def ptr_at(a, elem_offset):
    """void* to element `elem_offset` of a contiguous numpy array (for border[0], plane interior)."""
    return ctypes.c_void_p(a.ctypes.data + elem_offset * a.itemsize)


def pixel_dtype(bit_depth):
    return np.uint8 if bit_depth <= 8 else np.uint16
