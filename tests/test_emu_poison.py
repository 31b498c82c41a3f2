"""The SIMT interpreter's device memory is not zero (tests/simt_emu/simt_emu.cpp: every hipMalloc / hipHostMalloc is filled with 0xA5, as
fresh HBM and pinned memory hold whatever was there before): every emulated parity test of the CPU tier therefore also checks that no
kernel and no host path of the library relies on zero-filled allocations.  This pins the fill itself."""
import ctypes
import os

import pytest

from test_emu_picture import EMU_SO, emu_lib  # noqa: F401  (fixture)


@pytest.mark.skipif(bool(os.environ.get("SIMT_EMU_POISON")), reason="SIMT_EMU_POISON picks another fill")
def test_emulated_device_memory_is_not_zero(emu_lib):  # noqa: F811
    so = ctypes.CDLL(EMU_SO)
    malloc, free = so._Z9hipMallocPPvm, so._Z7hipFreePv
    malloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    free.argtypes = [ctypes.c_void_p]
    p = ctypes.c_void_p()
    assert malloc(ctypes.byref(p), 4096) == 0
    assert bytes((ctypes.c_ubyte * 4096).from_address(p.value)) == b"\xa5" * 4096
    free(p)
