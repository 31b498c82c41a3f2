"""Minimal ctypes driver of the reference's public C API (libde265/de265.h) — the dec265 decode loop
(dec265/dec265.cc:790-838) and its YUV writer (write_picture, dec265.cc:207-262) in Python, so that a test can decode a
bitstream through ANY libde265 build (the reference build oracle/_ref/libde265_ref.so, or glue/_build/libde265.so whose
pixel path is the MI355X backend) and hash what an application would see.  Test infrastructure."""
import ctypes
import hashlib

DE265_OK = 0
PARAM_BOOL_SEI_CHECK_HASH = 0
PARAM_ACCELERATION_CODE = 5
PARAM_DISABLE_DEBLOCKING = 7
PARAM_DISABLE_SAO = 8
ACCELERATION_SCALAR = 0


def bind(lib):
    vp, i = ctypes.c_void_p, ctypes.c_int
    lib.de265_new_decoder.restype = vp
    lib.de265_free_decoder.argtypes = [vp]
    lib.de265_start_worker_threads.argtypes = [vp, i]
    lib.de265_push_data.argtypes = [vp, vp, i, ctypes.c_int64, vp]
    lib.de265_flush_data.argtypes = [vp]
    lib.de265_decode.argtypes = [vp, ctypes.POINTER(i)]
    lib.de265_get_next_picture.argtypes = [vp]
    lib.de265_get_next_picture.restype = vp
    lib.de265_get_image_width.argtypes = [vp, i]
    lib.de265_get_image_height.argtypes = [vp, i]
    lib.de265_get_bits_per_pixel.argtypes = [vp, i]
    lib.de265_get_chroma_format.argtypes = [vp]
    lib.de265_get_image_plane.argtypes = [vp, i, ctypes.POINTER(i)]
    lib.de265_get_image_plane.restype = vp
    lib.de265_set_parameter_bool.argtypes = [vp, i, i]
    lib.de265_set_parameter_int.argtypes = [vp, i, i]
    lib.de265_get_warning.argtypes = [vp]
    return lib


def decode_stream(lib, data, threads=0, disable_deblocking=False, disable_sao=False, scalar=False, after_create=None, max_frames=None, check_hash=False,
                  planes_out=None, touch_planes=True, before_free=None):
    """-> (md5 hex of the output YUV, number of pictures, list of warnings).  The YUV is what `dec265 -o` writes: every
    output picture, cropped to the conformance window, planes Y Cb Cr, 8-bit samples as bytes / deeper ones as 2 bytes LE.
    planes_out: a list that receives every output picture's planes (numpy arrays); touch_planes=False: the pictures are taken
    from the decoder but their samples are never asked for (`dec265 -q` without -o)."""
    bind(lib)
    ctx = lib.de265_new_decoder()
    if not ctx:
        raise RuntimeError("de265_new_decoder failed")
    try:
        if scalar:
            lib.de265_set_parameter_int(ctx, PARAM_ACCELERATION_CODE, ACCELERATION_SCALAR)
        lib.de265_set_parameter_bool(ctx, PARAM_DISABLE_DEBLOCKING, int(disable_deblocking))
        lib.de265_set_parameter_bool(ctx, PARAM_DISABLE_SAO, int(disable_sao))
        lib.de265_set_parameter_bool(ctx, PARAM_BOOL_SEI_CHECK_HASH, int(check_hash))     # verify decoded-picture-hash SEIs (dec265 -c)
        if after_create:
            after_create(ctx)
        if threads > 0:
            assert lib.de265_start_worker_threads(ctx, threads) == DE265_OK
        buf = ctypes.create_string_buffer(data, len(data))
        assert lib.de265_push_data(ctx, buf, len(data), 0, None) == DE265_OK
        assert lib.de265_flush_data(ctx) == DE265_OK
        md5 = hashlib.md5()
        decode_errors = []
        n = 0
        more = ctypes.c_int(1)
        while more.value:
            more.value = 0
            err = lib.de265_decode(ctx, ctypes.byref(more))
            if err != DE265_OK:
                decode_errors.append(err)      # (a decoded-picture-hash mismatch, DE265_ERROR_CHECKSUM_MISMATCH = 5, ends the decode like dec265 -c)
                break
            while True:
                img = lib.de265_get_next_picture(ctx)
                if not img:
                    break
                nc = 1 if lib.de265_get_chroma_format(img) == 0 else 3
                pic_planes = []
                for c in range(nc if touch_planes else 0):
                    stride = ctypes.c_int()
                    p = lib.de265_get_image_plane(img, c, ctypes.byref(stride))
                    w, h = lib.de265_get_image_width(img, c), lib.de265_get_image_height(img, c)
                    bpp = (lib.de265_get_bits_per_pixel(img, c) + 7) // 8
                    row = ctypes.c_char * (w * bpp)
                    for y in range(h):
                        md5.update(row.from_address(p + y * stride.value))      # stride is in BYTES (de265.cc:735)
                    if planes_out is not None:
                        import numpy as np
                        a = np.empty((h, w), np.uint8 if bpp == 1 else np.uint16)
                        for y in range(h):
                            a[y] = np.frombuffer(row.from_address(p + y * stride.value), a.dtype, w)
                        pic_planes.append(a)
                if planes_out is not None:
                    planes_out.append(pic_planes)
                n += 1
                if max_frames and n >= max_frames:
                    more.value = 0
                    break
        warnings = []
        while True:
            wn = lib.de265_get_warning(ctx)
            if wn == DE265_OK:
                break
            warnings.append(wn)
        if before_free:
            before_free(ctx)
        return md5.hexdigest(), n, warnings + decode_errors
    finally:
        lib.de265_free_decoder(ctx)


class App:
    """One decoder driven the way applications drive it (dec265.cc:760-840, the reference's other front ends): data in pieces, NAL
    units one by one, pictures taken as they come — with get_next_picture or peek / release —, a reset in mid-stream (a seek),
    several decoders alive in one process.  result() = (md5 of everything shown, pictures shown, errors other than "waiting for
    input data")."""

    def __init__(self, lib, threads=0):
        self.lib = bind(lib)
        vp = ctypes.c_void_p
        lib.de265_peek_next_picture.argtypes = [vp]
        lib.de265_peek_next_picture.restype = vp
        lib.de265_release_next_picture.argtypes = [vp]
        lib.de265_reset.argtypes = [vp]
        lib.de265_push_NAL.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int64, vp]
        self.ctx = lib.de265_new_decoder()
        if not self.ctx:
            raise RuntimeError("de265_new_decoder failed")
        if threads:
            assert lib.de265_start_worker_threads(self.ctx, threads) == DE265_OK
        self.md5, self.n, self.errs = hashlib.md5(), 0, []

    def take(self, peek=False):
        lib = self.lib
        while True:
            img = lib.de265_peek_next_picture(self.ctx) if peek else lib.de265_get_next_picture(self.ctx)
            if not img:
                return
            for c in range(1 if lib.de265_get_chroma_format(img) == 0 else 3):
                stride = ctypes.c_int()
                p = lib.de265_get_image_plane(img, c, ctypes.byref(stride))
                w, h = lib.de265_get_image_width(img, c), lib.de265_get_image_height(img, c)
                row = ctypes.c_char * (w * ((lib.de265_get_bits_per_pixel(img, c) + 7) // 8))
                for y in range(h):
                    self.md5.update(row.from_address(p + y * stride.value))
            self.n += 1
            if peek:
                lib.de265_release_next_picture(self.ctx)

    def decode_some(self, peek=False):
        more = ctypes.c_int(1)
        while more.value:
            more.value = 0
            err = self.lib.de265_decode(self.ctx, ctypes.byref(more))
            if err != DE265_OK:
                if err != 13:                        # DE265_ERROR_WAITING_FOR_INPUT_DATA: push more
                    self.errs.append(err)
                break
            self.take(peek)

    def drain(self, peek=False):
        while True:
            before = self.n
            self.decode_some(peek)
            if self.n == before:
                return

    def push(self, b):
        buf = ctypes.create_string_buffer(b, len(b))
        assert self.lib.de265_push_data(self.ctx, buf, len(b), 0, None) == DE265_OK

    def push_nal(self, nal_with_start_code):
        b = nal_with_start_code[nal_with_start_code.index(b"\x00\x00\x01") + 3:]
        buf = ctypes.create_string_buffer(b, len(b))
        assert self.lib.de265_push_NAL(self.ctx, buf, len(b), 0, None) == DE265_OK

    def flush(self):
        assert self.lib.de265_flush_data(self.ctx) == DE265_OK

    def reset(self):
        self.lib.de265_reset(self.ctx)

    def close(self):
        if self.ctx:
            self.lib.de265_free_decoder(self.ctx)
            self.ctx = None

    def result(self):
        return self.md5.hexdigest(), self.n, self.errs
