"""The ways applications drive a decoder, through the glue (glue/_build/libde265.so, pixels from the backend) and through the reference
library, compared picture for picture: the bitstream pushed in pieces of 1 byte to 20 KB between decode calls (dec265.cc:760-840),
NAL units handed over one by one (de265_push_NAL), pictures taken with de265_peek_next_picture / de265_release_next_picture, a
de265_reset in mid-stream followed by the stream from its start (a seek: the decoder drops its DPB — the glue its frames, its queued
submits and the pictures it still owes), and two decoders of one process working on two streams in turns (two glue contexts on one
backend library).  tests/test_streams.py pushes every stream in one piece and only ever uses de265_get_next_picture."""
import random

import pytest

from de265_py import App
from libde265_amd import capi
from test_emu_picture import emu_lib, EMU_SO  # noqa: F401  (fixture)
from test_glue_live import glue_lib
from test_streams import F_RA, F_TMVP, F_WP, F_LT, make_stream, split_nals


def chunked(lib, data, seed, peek, threads):
    a = App(lib, threads)
    try:
        r, pos = random.Random(seed), 0
        while pos < len(data):
            n = r.choice([1, 7, 100, 4096, 20000])
            a.push(data[pos:pos + n])
            pos += n
            a.decode_some(peek)
        a.flush()
        a.drain(peek)
        return a.result()
    finally:
        a.close()


def by_nal(lib, data, threads):
    a = App(lib, threads)
    try:
        for _, b in split_nals(data):
            a.push_nal(b)
            a.decode_some()
        a.flush()
        a.drain()
        return a.result()
    finally:
        a.close()


def with_reset(lib, data, cut, threads):
    a = App(lib, threads)
    try:
        a.push(data[:cut])
        a.decode_some()
        a.reset()
        a.push(data)
        a.flush()
        a.drain()
        return a.result()
    finally:
        a.close()


def two_decoders(lib, d1, d2, threads):
    a, b = App(lib, threads), App(lib, threads)
    try:
        p1 = p2 = 0
        while p1 < len(d1) or p2 < len(d2):
            if p1 < len(d1):
                a.push(d1[p1:p1 + 3000]); p1 += 3000; a.decode_some()
            if p2 < len(d2):
                b.push(d2[p2:p2 + 5000]); p2 += 5000; b.decode_some(True)
        a.flush(); b.flush()
        a.drain(); b.drain(True)
        return a.result(), b.result()
    finally:
        a.close(); b.close()


def run_patterns(ref, lib, s1, s2, threads):
    for seed in (1, 2):
        for peek in (False, True):
            want = chunked(ref, s1, seed, peek, 0)
            assert want[1] > 0 and not want[2]
            assert chunked(lib, s1, seed, peek, threads) == want, "pieces, seed %d, peek %d" % (seed, peek)
    want = by_nal(ref, s1, 0)
    assert want[1] > 0 and by_nal(lib, s1, threads) == want, "NAL units one by one"
    want = with_reset(ref, s1, len(s1) // 2, 0)
    assert want[1] > 0 and with_reset(lib, s1, len(s1) // 2, threads) == want, "reset in mid-stream"
    want = two_decoders(ref, s1, s2, 0)
    assert want[0][1] > 0 and want[1][1] > 0 and two_decoders(lib, s1, s2, threads) == want, "two decoders in turns"
    assert lib.m355_glue_cpu_pixel_calls() == 0


@pytest.mark.parametrize("ra", [0, 1])
def test_application_patterns_emulated_backend(ref, emu_lib, tmp_path, monkeypatch, ra):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    s1 = make_stream(tmp_path, 256, 128, 8, 1, 1, 18, 403, 5, 1, 1, F_RA | F_TMVP) if ra else make_stream(tmp_path, 256, 128, 8, 1, 1, 8, 401, 10)
    s2 = make_stream(tmp_path, 192, 128, 10, 2, 1, 6, 402, 10, 1, 1, F_WP, 1, 2)
    run_patterns(ref, glue_lib(), s1, s2, 2)


@pytest.mark.gpu
def test_application_patterns_gpu(ref, tmp_path, monkeypatch):
    monkeypatch.delenv("M355_LIB", raising=False)
    s1 = make_stream(tmp_path, 1280, 720, 8, 1, 1, 18, 411, 5, 1, 1, F_RA | F_TMVP | F_LT)
    s2 = make_stream(tmp_path, 832, 480, 10, 2, 2, 6, 412, 10, 1, 1, F_WP, 1, 2)
    lib = glue_lib()
    run_patterns(ref, lib, s1, s2, 8)
    import os
    assert os.path.realpath(lib.m355_glue_backend_path().decode()) == os.path.realpath(capi.DEFAULT_LIB)
