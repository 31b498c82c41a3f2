"""The C-ABI boundary (include/de265_mi355x.h): the product library loads without a GPU and exports
every declared entry point; the POD sizes match the numpy mirrors; `struct m355_acceleration_functions`
has the size and landmark offsets of the REAL reference's `struct acceleration_functions`
(acceleration.h:29-231, obtained from the reference build through oracle/ref_shim.cc)."""
import ctypes
import os
import re
import subprocess

import numpy as np

from libde265_amd import capi, worklist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "de265_mi355x.h")


def declared_symbols():
    text = open(HEADER).read()
    text = text.replace("#define M355_API __attribute__((visibility(\"default\")))", "")
    return sorted(set(re.findall(r"M355_API\s+[^;(]*?\b(\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 20 and "init_acceleration_functions_mi355x" in names and "m355_submit_picture" in names
    out = subprocess.run(["nm", "-D", "--defined-only", capi.DEFAULT_LIB], check=True, capture_output=True, text=True).stdout
    exported = set(line.split()[-1] for line in out.splitlines() if line.strip())
    missing = [n for n in names if n not in exported]
    assert not missing, "declared in include/de265_mi355x.h but not exported: %s" % missing
    lib = capi.Library()           # binds every entry point; no device needed
    for n in names:
        assert getattr(lib.lib, n) is not None


def test_no_device_means_loud_failure_not_fallback():
    lib = capi.Library()
    if lib.device_count() > 0:
        return                      # (GPU box) covered by the gpu tier
    h = ctypes.c_void_p()
    assert lib.lib.m355_create(0, ctypes.byref(h)) == 1 and not h.value      # M355_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.lib.m355_last_error()
    t = capi.AccelerationFunctions()
    assert lib.lib.init_acceleration_functions_mi355x(ctypes.byref(t)) == 1
    assert bytes(t) == bytes(ctypes.sizeof(t))                               # table untouched
    a = np.zeros(64, np.uint8); off = np.zeros(1, np.int64); c = np.zeros(16, np.int16)
    assert lib.lib.m355_transform_add_batch(1, 2, 0, 8, a.ctypes.data, ctypes.c_size_t(64), off.ctypes.data,
                                            ctypes.c_ssize_t(8), c.ctypes.data) == 1


def test_pod_sizes_match_header():
    text = open(HEADER).read()
    sizes = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"\}\s*(m355_\w+);\s*/\*\s*(\d+) bytes", text))
    mirrors = {"m355_pic_params": worklist.PIC_PARAMS, "m355_slice": worklist.SLICE, "m355_ctb": worklist.CTB,
               "m355_cu": worklist.CU, "m355_tu": worklist.TU, "m355_wt": worklist.WT, "m355_pb": worklist.PB,
               "m355_rb": worklist.RB, "m355_ib": worklist.IB}
    assert set(sizes) == set(mirrors)
    for k, dt in mirrors.items():
        assert dt.itemsize == sizes[k], k
    # and as the C compiler sees them
    src = "#include <stdio.h>\n#include \"de265_mi355x.h\"\nint main(){printf(\"%s\\n\"" % " ".join(["%zu"] * (len(mirrors) + 2))
    src += "".join(", sizeof(%s)" % k for k in mirrors) + ", sizeof(m355_picture), sizeof(struct m355_acceleration_functions)); return 0;}"
    exe = os.path.join(ROOT, ".pytest_cache", "abi_sizes")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src, text=True, check=True)
    got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert got[:len(mirrors)] == [dt.itemsize for dt in mirrors.values()]
    assert got[-2] == ctypes.sizeof(worklist.CPicture)
    assert got[-1] == ctypes.sizeof(capi.AccelerationFunctions) == 752


def test_slot_table_layout_matches_reference(ref):
    ref.ref_accel_sizeof.restype = ctypes.c_size_t
    ref.ref_accel_offsetof.restype = ctypes.c_size_t
    A = capi.AccelerationFunctions
    assert ref.ref_accel_sizeof() == ctypes.sizeof(A)
    landmarks = ["put_weighted_pred_avg_8", "put_hevc_epel_8", "put_hevc_qpel_8", "put_hevc_qpel_16", "transform_bypass",
                 "transform_add_8", "transform_add_16", "add_residual_8", "dequant_coeff_block", "deblock_luma_8",
                 "rdpcm_v", "intra_pred_dc_8", "intra_pred_angular_16", "fwd_transform_4x4_dst_8", "hadamard_transform_8"]
    for i, name in enumerate(landmarks):
        assert ref.ref_accel_offsetof(i) == getattr(A, name).offset, name
