"""Synthetic HEVC bitstreams (oracle/ref_streamgen.cc: our coding-tree syntax writer on the reference's own CABAC encoder,
parameter-set / slice-header writers and residual writer) decoded LIVE through de265.h twice — by the reference decoder
(oracle/_ref/libde265_ref.so, scalar and SIMD) and by glue/_build/libde265.so, whose pixels come from the MI355X backend —
and compared bit for bit.  This is what pins the glue's recorder and the kernels on what girlshy does not contain: P and B
slices with merge / skip / AMVP motion the DECODER derives, two reference pictures, uniform tiles decoded by parallel threads,
10-bit samples, AMP partitions, intra NxN inside inter pictures, SAO parameters with merge candidates — and, through the
generator's own writers (feature streams below), explicit weighted prediction, transform skip, RDPCM, transform_skip_rotation,
cu_transquant_bypass, cu_qp_delta / slice QP and chroma QP offsets, PCM units, default and explicit scaling lists, constrained
intra prediction, several (dependent) slice segments per picture with their own deblocking / SAO parameters, monochrome /
4:2:2 / 4:4:4 with cross-component prediction; CTBs of 32 and 16, coding blocks from 16x16 (inter NxN), non-uniform tiles, no filtering
across tile boundaries, parallel merge level, a conformance window (geometry cases).  Each feature case also asserts, through the glue's coverage counters, that the
recorder branch which maps the feature really ran.

CPU tier: small streams, backend = SIMT-interpreter build.  GPU tier: up to 4K tiled 10-bit."""
import ctypes
import hashlib
import os
import subprocess

import pytest

import de265_py
from libde265_amd import capi
from test_emu_picture import emu_lib, EMU_SO  # noqa: F401  (fixture)
from test_glue_live import glue_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMGEN = os.path.join(ROOT, "oracle", "_ref", "streamgen")


# feature bits of oracle/ref_streamgen.cc (what the reference's own writers cannot express: written by OUR writers there)
F_WP, F_TSKIP, F_BYPASS, F_QPDELTA, F_PCM, F_SCALING, F_SCALING_PPS, F_REXT, F_CIP, F_DEPSLICE = 1, 2, 4, 8, 16, 32, 64, 256, 512, 1024
# random-access shape (the offline stand-in for the ra_main conformance streams): hierarchical-B groups of 8 decoded out of output
# order, slice-header reference picture sets with up to 6 pictures and 4 active references per list, a long-term picture,
# temporal motion vector prediction, sign data hiding, and WPP substreams (one per CTB row, parsed by the reference's WPP threads)
F_RA, F_WPP, F_TMVP, F_SDH, F_LT = 2048, 4096, 8192, 16384, 32768
# slice headers that differ inside a picture (slice type, active references, merge candidates, cabac_init_flag), reference picture list
# modification (lists naming a picture twice / in another order), pictures that are decoded and referenced but never shown
F_MIXSLICE, F_LISTMOD, F_NOOUTPUT = 65536, 131072, 262144
# geometry bits of oracle/ref_streamgen.cc (`geom`): CTB 32 / 16 instead of 64, coding blocks of at least 16x16 (inter NxN), explicit
# (non-uniform) tile column widths / row heights, no in-loop filtering across tile boundaries, log2_parallel_merge_level 4, TBs of at
# most 16x16 + strong intra smoothing off, a conformance window (the application sees a cropped picture)
G_CTB32, G_CTB16, G_MINCB16, G_TILES, G_NOTILEFILTER, G_PARMERGE, G_TB16, G_CONFWIN = 1, 2, 4, 8, 16, 32, 64, 128
# coverage counters of the glue's recorder (glue/m355_glue.cc FEAT_*): a stream that carries a feature must drive its branch
FEATS = ["pcm_cu", "weighted_pb", "bypass_rb", "skip_rb", "rdpcm_rb", "rotate_rb", "scaling_rb", "cross_comp_rb", "multi_slice_pic",
         "weighted_pb_later_slice", "no_boundary_filter_ib", "fill_pb", "chroma_422_rb", "chroma_444_rb", "mono_pic", "deblock_off_slice"]


def make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, intra_pct=5, b_frames=1, sao=1, features=0, chroma=1, slices=1, geom=0):
    if not os.path.exists(STREAMGEN):
        if not os.path.isdir("/root/reference"):
            pytest.skip("oracle/_ref/streamgen not available here")
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-j8", "gen"], check=True, stdout=subprocess.DEVNULL)
    out = os.path.join(str(tmp_path), "s_%dx%d_%d_%dx%d_%d_%d.h265" % (w, h, bd, tc, tr, seed, geom))
    subprocess.run([STREAMGEN, out, str(w), str(h), str(bd), str(tc), str(tr), str(frames), str(seed), str(intra_pct), str(b_frames), str(sao),
                    str(features), str(chroma), str(slices), str(geom)], check=True)
    return open(out, "rb").read()


def shown_pictures(frames, feat):
    """pictures the decoder hands out: all of them, minus (F_NOOUTPUT) those the writer gives pic_output_flag = 0 (POC % 3 == 1)"""
    if not feat & F_NOOUTPUT:
        return frames
    if feat & F_RA:
        pocs = [0]
        base = 0
        while len(pocs) < frames:
            pocs += [base + g for g in (8, 4, 2, 1, 3, 6, 5, 7)]
            base += 8
        pocs = pocs[:frames]
    else:
        pocs = list(range(frames))
    return sum(1 for i, p in enumerate(pocs) if i == 0 or p % 3 != 1)


def feature_counts(lib):
    lib.m355_glue_feature_counts.argtypes = [ctypes.POINTER(ctypes.c_longlong), ctypes.c_int]
    buf = (ctypes.c_longlong * len(FEATS))()
    assert lib.m355_glue_feature_counts(buf, len(FEATS)) == len(FEATS)
    return dict(zip(FEATS, list(buf)))


def check(ref, data, frames, threads, backend, expect=()):
    want = de265_py.decode_stream(ref, data, threads=0, scalar=True)
    assert want[1] == frames and not want[2], "the reference itself rejects the generated stream: %r" % (want,)
    assert de265_py.decode_stream(ref, data, threads=threads)[0] == want[0]          # SIMD tables, tile threads
    lib = glue_lib()
    before = feature_counts(lib)
    got = de265_py.decode_stream(lib, data, threads=threads)
    assert got[:2] == want[:2], "live decode on the backend differs from the reference decoder"
    assert set(got[2]) <= {1000}, got[2]       # (DE265_WARNING_NO_WPP_CANNOT_USE_MULTITHREADING: threads asked for on a stream without tiles)
    assert lib.m355_glue_cpu_pixel_calls() == 0
    assert os.path.realpath(lib.m355_glue_backend_path().decode()) == os.path.realpath(backend)
    after = feature_counts(lib)
    for name in expect:                         # the stream really drove the recorder branch it is there for
        assert after[name] > before[name], "the stream did not exercise '%s' (%r)" % (name, {k: after[k] - before[k] for k in FEATS})
    return want


def check_random_access(ref, data, frames, threads, backend):
    """A random-access stream: as check(), plus the decode an application makes that never looks at the samples (dec265 -q
    without -o: pictures are taken in output order and dropped, frames recycle while the submit worker is still behind)."""
    want = check(ref, data, frames, threads, backend)
    lib = glue_lib()
    got = de265_py.decode_stream(lib, data, threads=threads, touch_planes=False)
    assert got[1] == frames and set(got[2]) <= {1000}, got
    assert lib.m355_glue_cpu_pixel_calls() == 0
    # and once more with the samples, single-threaded: the glue's frame bookkeeping must not depend on the parser's threads
    assert de265_py.decode_stream(lib, data, threads=0)[:2] == want[:2]


CPU_CASES = [(416, 240, 8, 1, 1, 4, 21), (448, 256, 10, 2, 2, 4, 22)]


@pytest.mark.parametrize("w,h,bd,tc,tr,frames,seed", CPU_CASES)
def test_generated_streams_emulated_backend(ref, emu_lib, tmp_path, monkeypatch, w, h, bd, tc, tr, frames, seed):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    check(ref, make_stream(tmp_path, w, h, bd, tc, tr, frames, seed), frames, 4, EMU_SO)


# Feature streams: everything the glue's recorder maps by hand (glue/m355_glue.cc), each pinned against the reference DECODER.
# (w, h, bit depth, tile cols, tile rows, frames, seed, intra %, features, chroma format, slices, recorder branches that must run)
FEATURE_CPU_CASES = [
    (256, 128, 8, 1, 1, 3, 41, 20, F_WP | F_TSKIP | F_BYPASS | F_QPDELTA | F_PCM | F_SCALING_PPS | F_REXT | F_CIP | F_DEPSLICE, 1, 3,
     ("pcm_cu", "weighted_pb", "bypass_rb", "skip_rb", "rdpcm_rb", "scaling_rb", "multi_slice_pic", "weighted_pb_later_slice")),
    (192, 128, 8, 1, 1, 3, 42, 30, F_REXT | F_TSKIP | F_BYPASS, 3, 1, ("cross_comp_rb", "chroma_444_rb", "rotate_rb", "rdpcm_rb", "no_boundary_filter_ib")),
    (192, 128, 10, 2, 1, 3, 43, 30, F_REXT | F_TSKIP | F_PCM | F_WP, 2, 2, ("chroma_422_rb", "pcm_cu", "weighted_pb", "multi_slice_pic")),
    (192, 128, 8, 1, 1, 3, 44, 10, F_SCALING | F_QPDELTA, 0, 2, ("mono_pic", "scaling_rb")),
]


@pytest.mark.parametrize("w,h,bd,tc,tr,frames,seed,intra,feat,chroma,slices,expect", FEATURE_CPU_CASES)
def test_feature_streams_emulated_backend(ref, emu_lib, tmp_path, monkeypatch, w, h, bd, tc, tr, frames, seed, intra, feat, chroma, slices, expect):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    check(ref, make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, intra, 1, 1, feat, chroma, slices), frames, 3, EMU_SO, expect)


# (w, h, bit depth, frames, seed, features, parser threads)
RA_CPU_CASES = [(256, 128, 8, 14, 5, F_RA | F_LT | F_TMVP | F_SDH, 3), (256, 192, 8, 10, 6, F_RA | F_WPP | F_TMVP | F_SDH, 4)]


@pytest.mark.parametrize("w,h,bd,frames,seed,feat,threads", RA_CPU_CASES)
def test_random_access_streams_emulated_backend(ref, emu_lib, tmp_path, monkeypatch, w, h, bd, frames, seed, feat, threads):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    check_random_access(ref, make_stream(tmp_path, w, h, bd, 1, 1, frames, seed, 5, 1, 1, feat), frames, threads, EMU_SO)


RA_GPU_CASES = [
    (3840, 2160, 8, 17, 81, F_RA | F_LT | F_TMVP | F_SDH, 8),              # the C3 stand-in: 4K, two groups of pictures + the IDR
    (3840, 2160, 8, 17, 82, F_RA | F_LT | F_TMVP | F_SDH | F_WPP, 8),      # ... WPP-only, CTB rows on 8 parser threads
    (1920, 1080, 10, 33, 83, F_RA | F_LT | F_TMVP | F_SDH | F_WPP | F_WP, 8),   # four groups: the long-term picture outlives three of them
    (832, 480, 8, 41, 84, F_RA | F_TMVP, 0),
]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bd,frames,seed,feat,threads", RA_GPU_CASES)
def test_random_access_streams_gpu(ref, tmp_path, monkeypatch, w, h, bd, frames, seed, feat, threads):
    monkeypatch.delenv("M355_LIB", raising=False)
    check_random_access(ref, make_stream(tmp_path, w, h, bd, 1, 1, frames, seed, 5, 1, 1, feat), frames, threads, capi.DEFAULT_LIB)


# Picture / block geometry away from the generator's default (CTB 64, CBs from 8x8, uniform tiles, whole picture shown): what real
# streams vary and the cases above do not.
# (w, h, bit depth, tile cols, tile rows, frames, seed, intra %, features, chroma format, slices, geometry, parser threads, ranks)
GEOM_CPU_CASES = [
    (256, 128, 8, 1, 1, 3, 101, 10, 0, 1, 1, G_CTB32, 2, 0),
    (256, 128, 8, 2, 2, 3, 104, 10, 0, 1, 1, G_CTB16 | G_TILES | G_NOTILEFILTER, 4, 0),
    (256, 128, 8, 1, 1, 3, 106, 10, 0, 1, 1, G_CTB32 | G_MINCB16 | G_PARMERGE, 2, 0),
    (192, 128, 8, 1, 1, 3, 110, 10, 0, 3, 2, G_CTB16 | G_TB16 | G_CONFWIN, 2, 0),
    (192, 128, 10, 2, 1, 4, 115, 20, F_WP | F_TSKIP | F_QPDELTA | F_PCM | F_REXT, 3, 2, G_CTB32 | G_MINCB16 | G_TILES | G_TB16, 3, 0),
    (256, 192, 8, 1, 1, 10, 113, 5, F_RA | F_WPP | F_TMVP | F_SDH, 1, 1, G_CTB32, 4, 0),
    (256, 128, 8, 1, 1, 10, 114, 5, F_RA | F_LT | F_TMVP | F_SDH, 1, 1, G_CTB16 | G_PARMERGE | G_CONFWIN, 3, 0),
    (320, 192, 8, 3, 2, 4, 119, 15, F_WP | F_QPDELTA | F_PCM, 1, 1, G_CTB16 | G_TILES | G_NOTILEFILTER, 4, 3),   # uneven tiles over three ranks
]
GEOM_GPU_CASES = [
    (832, 480, 8, 1, 1, 5, 131, 10, 0, 1, 1, G_CTB32, 8, 0),
    (832, 480, 10, 1, 1, 5, 132, 10, F_QPDELTA, 1, 2, G_CTB16, 8, 0),
    (1280, 720, 8, 3, 2, 4, 133, 10, 0, 1, 1, G_TILES | G_NOTILEFILTER, 8, 0),
    (1920, 1088, 8, 2, 2, 4, 134, 10, F_WP | F_QPDELTA, 1, 3, G_CTB32 | G_TILES | G_CONFWIN, 8, 0),              # 1080 lines shown out of 1088 coded
    (832, 480, 8, 1, 1, 5, 135, 20, F_PCM | F_TSKIP | F_REXT, 3, 1, G_CTB16 | G_MINCB16 | G_TB16, 8, 0),
    (1920, 1088, 8, 1, 1, 17, 136, 5, F_RA | F_LT | F_TMVP | F_SDH | F_WPP, 1, 1, G_CTB32 | G_PARMERGE | G_CONFWIN, 8, 0),
    (1920, 1088, 8, 3, 2, 4, 137, 10, F_WP, 1, 1, G_CTB32 | G_TILES, 8, 3),
]


# Slice headers: what one picture's slices may disagree on, reordered reference lists, pictures that are never shown (same case layout)
HEADER_CPU_CASES = [
    (256, 128, 8, 1, 1, 6, 301, 10, F_MIXSLICE, 1, 4, 0, 2, 0),
    (256, 128, 8, 2, 2, 6, 302, 15, F_MIXSLICE | F_WP | F_QPDELTA | F_DEPSLICE, 1, 5, 0, 4, 0),
    (256, 128, 8, 1, 1, 6, 304, 10, F_LISTMOD | F_MIXSLICE | F_WP, 1, 3, 0, 2, 0),
    (256, 128, 8, 1, 1, 26, 308, 5, F_RA | F_TMVP | F_LT | F_SDH | F_LISTMOD | F_NOOUTPUT | F_MIXSLICE, 1, 1, G_CTB16, 3, 0),
    (320, 192, 8, 2, 2, 7, 309, 15, F_MIXSLICE | F_LISTMOD | F_NOOUTPUT | F_WP | F_QPDELTA | F_PCM | F_CIP | F_DEPSLICE, 1, 5, G_CTB32 | G_TILES, 4, 2),
]
HEADER_GPU_CASES = [
    (1280, 720, 8, 2, 2, 6, 321, 10, F_MIXSLICE | F_LISTMOD | F_WP | F_QPDELTA, 1, 6, 0, 8, 0),
    (1920, 1080, 8, 1, 1, 25, 322, 5, F_RA | F_TMVP | F_LT | F_SDH | F_WPP | F_LISTMOD | F_NOOUTPUT | F_MIXSLICE, 1, 1, 0, 8, 0),
    (832, 480, 10, 1, 1, 8, 323, 10, F_NOOUTPUT | F_LISTMOD, 1, 1, G_CTB32, 8, 0),
]


def check_geometry(ref, tmp_path, monkeypatch, case, backend):
    w, h, bd, tc, tr, frames, seed, intra, feat, chroma, slices, geom, threads, ranks = case
    if ranks:
        monkeypatch.setenv("M355_GLUE_RANKS", str(ranks))
    data = make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, intra, 1, 1, feat, chroma, slices, geom)
    (check_random_access if feat & F_RA else check)(ref, data, shown_pictures(frames, feat), threads, backend)


@pytest.mark.parametrize("case", GEOM_CPU_CASES, ids=lambda c: "%dx%d-seed%d-geom%d" % (c[0], c[1], c[6], c[11]))
def test_geometry_streams_emulated_backend(ref, emu_lib, tmp_path, monkeypatch, case):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    check_geometry(ref, tmp_path, monkeypatch, case, EMU_SO)


@pytest.mark.parametrize("case", HEADER_CPU_CASES, ids=lambda c: "%dx%d-seed%d-feat%d" % (c[0], c[1], c[6], c[8]))
def test_slice_header_streams_emulated_backend(ref, emu_lib, tmp_path, monkeypatch, case):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    check_geometry(ref, tmp_path, monkeypatch, case, EMU_SO)


@pytest.mark.gpu
@pytest.mark.parametrize("case", HEADER_GPU_CASES, ids=lambda c: "%dx%d-seed%d-feat%d" % (c[0], c[1], c[6], c[8]))
def test_slice_header_streams_gpu(ref, tmp_path, monkeypatch, case):
    monkeypatch.delenv("M355_LIB", raising=False)
    check_geometry(ref, tmp_path, monkeypatch, case, capi.DEFAULT_LIB)


@pytest.mark.gpu
@pytest.mark.parametrize("case", GEOM_GPU_CASES, ids=lambda c: "%dx%d-seed%d-geom%d" % (c[0], c[1], c[6], c[11]))
def test_geometry_streams_gpu(ref, tmp_path, monkeypatch, case):
    monkeypatch.delenv("M355_LIB", raising=False)
    check_geometry(ref, tmp_path, monkeypatch, case, capi.DEFAULT_LIB)


# Damaged streams.  A picture that never arrives: the decoder makes up the reference pictures the later ones name (decctx.cc
# generate_unavailable_reference_picture: mid-grey planes in HOST memory — the glue has to carry them to the device as reference
# frames, glue/m355_glue.cc upload_host_planes) and carries on; flipped bits inside slice data: the parser stops somewhere in the
# picture (DE265_WARNING_CTB_OUTSIDE_IMAGE_AREA and friends), the CTBs it never reached keep what the picture buffer held.  The backend
# must show what the reference decoder shows, sample for sample, and raise the same warnings.
def split_nals(data):
    """[(nal_unit_type, bytes incl. start code)]"""
    import re
    pos = [m.start() for m in re.finditer(b"\x00\x00\x01", data)]
    out = []
    for i, p in enumerate(pos):
        s = p - 1 if p > 0 and data[p - 1] == 0 else p
        e = len(data) if i + 1 == len(pos) else (pos[i + 1] - 1 if data[pos[i + 1] - 1] == 0 else pos[i + 1])
        out.append(((data[p + 3] >> 1) & 0x3F, data[s:e]))
    return out


def drop_pictures(data, pics):
    """the stream without the slice NAL units of the pictures whose coding index is in `pics`"""
    out, pic = [], -1
    for t, b in split_nals(data):
        if t < 32:
            if b[b.index(b"\x00\x00\x01") + 5] & 0x80:      # first_slice_segment_in_pic_flag
                pic += 1
            if pic in pics:
                continue
        out.append(b)
    return b"".join(out)


def flip_bits(data, seed):
    import random
    rnd = random.Random(seed)
    out = []
    for t, b in split_nals(data):
        if t < 32 and len(b) > 40 and rnd.random() < 0.5:
            b = bytearray(b)
            for _ in range(rnd.randint(1, 4)):
                b[rnd.randint(12, len(b) - 1)] ^= 1 << rnd.randint(0, 7)
            b = bytes(b)
        out.append(b)
    return b"".join(out)


def check_damaged(ref, data, threads, backend):
    want = de265_py.decode_stream(ref, data, threads=0, scalar=True)
    lib = glue_lib()
    got = de265_py.decode_stream(lib, data, threads=threads)
    assert got[:2] == want[:2], "damaged stream: the backend shows something else than the reference decoder"
    assert set(got[2]) - {1000} == set(want[2]) - {1000}, (got[2], want[2])
    assert lib.m355_glue_cpu_pixel_calls() == 0
    assert os.path.realpath(lib.m355_glue_backend_path().decode()) == os.path.realpath(backend)
    # DE265_DECODER_PARAM_SUPPRESS_FAULTY_PICTURES (de265.h:404, decctx.cc:1854): only pictures whose integrity is "correct" are
    # shown — the marks the replaced functions own (motion.cc:387-405 and friends) have to be where the decoder looks for them
    shown = [de265_py.decode_stream(x, data, threads=t, scalar=x is ref, after_create=lambda ctx, x=x: x.de265_set_parameter_bool(ctx, 6, 1))
             for x, t in ((ref, 0), (lib, threads))]
    assert shown[1][:2] == shown[0][:2] and shown[0][1] <= want[1], "faulty pictures suppressed: %r, reference %r" % (shown[1][1:], shown[0][1:])
    return want


# (stream arguments of make_stream, pictures to drop (coding order))
DROP_CPU_CASES = [((256, 128, 8, 1, 1, 8, 201, 10), [2]), ((256, 128, 8, 1, 1, 8, 201, 10), [0]), ((256, 128, 8, 1, 1, 8, 201, 10), [3, 4]),
                  ((256, 128, 8, 1, 1, 18, 202, 5, 1, 1, F_RA | F_TMVP), [1]), ((256, 128, 8, 1, 1, 18, 202, 5, 1, 1, F_RA | F_TMVP), [1, 2, 3, 9])]
DROP_GPU_CASES = [((832, 480, 8, 2, 1, 8, 211, 10), [2]), ((1280, 720, 10, 1, 1, 18, 212, 5, 1, 1, F_RA | F_TMVP | F_LT), [1, 2, 9]), ((832, 480, 8, 1, 1, 6, 213, 10), [0])]


@pytest.mark.parametrize("args,pics", DROP_CPU_CASES)
def test_lost_pictures_emulated_backend(ref, emu_lib, tmp_path, monkeypatch, args, pics):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    data = make_stream(tmp_path, *args)
    want = check_damaged(ref, drop_pictures(data, set(pics)), 2, EMU_SO)
    assert want[1] == args[5] - len(pics)                # every picture that arrived is shown


@pytest.mark.parametrize("seed", range(6))
def test_flipped_bits_emulated_backend(ref, emu_lib, tmp_path, monkeypatch, seed):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    data = make_stream(tmp_path, 256, 128, 8, 1, 1, 8, 201, 10)
    check_damaged(ref, flip_bits(data, seed), 0, EMU_SO)


@pytest.mark.gpu
@pytest.mark.parametrize("args,pics", DROP_GPU_CASES)
def test_lost_pictures_gpu(ref, tmp_path, monkeypatch, args, pics):
    monkeypatch.delenv("M355_LIB", raising=False)
    data = make_stream(tmp_path, *args)
    want = check_damaged(ref, drop_pictures(data, set(pics)), 8, capi.DEFAULT_LIB)
    assert want[1] == args[5] - len(pics)


# One bitstream across several backend contexts of ONE process (M355_GLUE_RANKS: the glue splits every picture's lists by tile
# owner, the library's m355_group_* carries the halos and the finished tiles between the contexts; glue/m355_glue.cc submit_sharded):
# (w, h, bit depth, tile cols, tile rows, frames, seed, features, ranks)
RANKS_CPU_CASES = [(448, 256, 10, 2, 2, 4, 22, 0, 2), (320, 192, 8, 3, 2, 4, 23, F_WP | F_QPDELTA | F_PCM, 3)]


@pytest.mark.parametrize("w,h,bd,tc,tr,frames,seed,feat,ranks", RANKS_CPU_CASES)
def test_one_stream_across_ranks_emulated_backend(ref, emu_lib, tmp_path, monkeypatch, w, h, bd, tc, tr, frames, seed, feat, ranks):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    monkeypatch.setenv("M355_GLUE_RANKS", str(ranks))
    check(ref, make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, 10, 1, 1, feat), frames, 4, EMU_SO)


RANKS_GPU_CASES = [(3840, 2160, 10, 2, 2, 5, 85, 0, 4), (1920, 1080, 8, 4, 2, 6, 86, F_WP | F_QPDELTA | F_PCM | F_CIP, 8),
                   (1920, 1080, 8, 3, 2, 17, 87, F_RA | F_LT | F_TMVP | F_SDH, 3), (7680, 4320, 10, 4, 2, 3, 88, 0, 8)]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bd,tc,tr,frames,seed,feat,ranks", RANKS_GPU_CASES)
def test_one_stream_across_ranks_gpu(ref, tmp_path, monkeypatch, w, h, bd, tc, tr, frames, seed, feat, ranks):
    """all ranks share the one GPU of the test box (contexts are per rank, the device index wraps): the control flow, the list
    split and the exchanges are those of a node with one GPU per rank"""
    monkeypatch.delenv("M355_LIB", raising=False)
    monkeypatch.setenv("M355_GLUE_RANKS", str(ranks))
    check(ref, make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, 10, 1, 1, feat), frames, 8, capi.DEFAULT_LIB)


FEATURE_GPU_CASES = [
    (832, 480, 8, 1, 1, 6, 51, 5, F_WP, 1, 1, ("weighted_pb",)),
    (832, 480, 10, 2, 2, 6, 52, 10, F_WP | F_QPDELTA, 1, 4, ("weighted_pb", "weighted_pb_later_slice", "multi_slice_pic")),
    (832, 480, 8, 1, 1, 5, 53, 10, F_TSKIP, 1, 1, ("skip_rb",)),
    (832, 480, 8, 1, 1, 5, 54, 15, F_BYPASS, 1, 1, ("bypass_rb",)),
    (832, 480, 9, 1, 1, 5, 55, 10, F_QPDELTA, 1, 3, ("multi_slice_pic",)),
    (832, 480, 8, 2, 1, 5, 56, 40, F_PCM, 1, 1, ("pcm_cu",)),
    (832, 480, 10, 1, 1, 5, 57, 40, F_PCM | F_CIP, 1, 2, ("pcm_cu",)),
    (832, 480, 8, 1, 1, 5, 58, 10, F_SCALING, 1, 1, ("scaling_rb",)),
    (832, 480, 10, 1, 1, 5, 59, 10, F_SCALING_PPS, 1, 1, ("scaling_rb",)),
    (832, 480, 8, 1, 1, 5, 60, 30, F_CIP, 1, 1, ()),
    (1280, 720, 8, 1, 1, 4, 61, 5, 0, 1, 6, ("multi_slice_pic", "deblock_off_slice")),
    (1280, 720, 8, 3, 2, 4, 62, 5, F_DEPSLICE, 1, 5, ("multi_slice_pic",)),
    (832, 480, 8, 1, 1, 5, 63, 10, F_REXT | F_TSKIP | F_BYPASS, 1, 1, ("rdpcm_rb", "rotate_rb", "skip_rb", "bypass_rb")),
    (832, 480, 8, 1, 1, 5, 64, 10, 0, 2, 1, ("chroma_422_rb",)),
    (832, 480, 8, 1, 1, 5, 65, 10, 0, 3, 1, ("chroma_444_rb",)),
    (832, 480, 8, 1, 1, 5, 66, 10, 0, 0, 1, ("mono_pic",)),
    (832, 480, 10, 2, 1, 5, 67, 30, F_REXT | F_TSKIP | F_BYPASS | F_WP | F_QPDELTA, 3, 2, ("cross_comp_rb", "chroma_444_rb", "rdpcm_rb", "weighted_pb")),
    (832, 480, 12, 1, 2, 5, 68, 30, F_REXT | F_TSKIP | F_BYPASS | F_PCM | F_SCALING_PPS, 2, 2, ("chroma_422_rb", "pcm_cu", "scaling_rb")),
    (1920, 1080, 8, 2, 2, 4, 69, 20, F_WP | F_TSKIP | F_BYPASS | F_QPDELTA | F_PCM | F_SCALING_PPS | F_REXT | F_CIP | F_DEPSLICE, 1, 6,
     ("pcm_cu", "weighted_pb", "bypass_rb", "skip_rb", "rdpcm_rb", "scaling_rb", "multi_slice_pic", "weighted_pb_later_slice")),
    (1920, 1080, 10, 1, 1, 3, 70, 100, F_TSKIP | F_BYPASS | F_QPDELTA | F_PCM | F_SCALING | F_REXT | F_CIP, 3, 3, ("pcm_cu", "cross_comp_rb", "scaling_rb")),
]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bd,tc,tr,frames,seed,intra,feat,chroma,slices,expect", FEATURE_GPU_CASES)
def test_feature_streams_gpu(ref, tmp_path, monkeypatch, w, h, bd, tc, tr, frames, seed, intra, feat, chroma, slices, expect):
    monkeypatch.delenv("M355_LIB", raising=False)
    check(ref, make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, intra, 1, 1, feat, chroma, slices), frames, 8, capi.DEFAULT_LIB, expect)


GPU_CASES = [(416, 240, 8, 1, 1, 8, 31, 5, 1, 1), (832, 480, 10, 3, 2, 8, 32, 10, 1, 1), (1920, 1080, 8, 2, 1, 5, 33, 30, 0, 1),
             (1280, 720, 12, 2, 2, 5, 34, 5, 1, 0), (3840, 2160, 10, 2, 2, 3, 35, 5, 1, 1), (640, 368, 8, 1, 1, 6, 36, 100, 0, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bd,tc,tr,frames,seed,intra,b,sao", GPU_CASES)
def test_generated_streams_gpu(ref, tmp_path, monkeypatch, w, h, bd, tc, tr, frames, seed, intra, b, sao):
    monkeypatch.delenv("M355_LIB", raising=False)
    check(ref, make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, intra, b, sao), frames, 8, capi.DEFAULT_LIB)


# New coded video sequences inside one bitstream: every segment starts with its own VPS / SPS / PPS and an IDR picture — other
# picture size, bit depth, chroma format and tile structure than the one before.  The decoder (decctx.cc:1303-1356 process_sps /
# image allocation dpb.cc:239-281) re-creates its images; the glue's frames, the lanes' working planes and scratch, the scan-table
# cache and the pinned picture allocator have to follow.
def segments(tmp_path, specs):
    return b"".join(make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, 10, 1, 1, feat, chroma) for (w, h, bd, tc, tr, frames, seed, feat, chroma) in specs)


SEGMENTS_CPU = [(256, 128, 8, 1, 1, 3, 71, 0, 1), (192, 192, 10, 2, 1, 3, 72, 0, 1), (128, 64, 8, 1, 1, 2, 73, 0, 3), (256, 128, 8, 1, 1, 3, 71, F_WP, 1)]
SEGMENTS_GPU = [(1920, 1080, 8, 2, 2, 4, 91, 0, 1), (3840, 2160, 10, 4, 2, 3, 92, 0, 1), (832, 480, 8, 1, 1, 4, 93, F_TSKIP, 3),
                (1280, 720, 12, 1, 1, 3, 94, 0, 2), (1920, 1080, 8, 2, 2, 4, 91, F_WP | F_QPDELTA, 1)]


def test_parameter_sets_change_mid_stream_emulated_backend(ref, emu_lib, tmp_path, monkeypatch):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    check(ref, segments(tmp_path, SEGMENTS_CPU), sum(s[5] for s in SEGMENTS_CPU), 2, EMU_SO)


@pytest.mark.gpu
def test_parameter_sets_change_mid_stream_gpu(ref, tmp_path, monkeypatch):
    monkeypatch.delenv("M355_LIB", raising=False)
    data = segments(tmp_path, SEGMENTS_GPU)
    n = sum(s[5] for s in SEGMENTS_GPU)
    check(ref, data, n, 8, capi.DEFAULT_LIB)
    # ... and with every picture taken without a look at its samples (frames of the old size are released while decodes of the new size run)
    got = de265_py.decode_stream(glue_lib(), data, threads=8, touch_planes=False)
    assert got[1] == n and set(got[2]) <= {1000}, got


# transform_skip_rotation of a CHROMA block: the reference looks the prediction mode up at the block's chroma coordinates in the luma-indexed array
# (transform.cc:398) — with 4:2:2 and tile columns a position in ANOTHER tile, which that tile's thread may or may not have parsed yet: the reference's own
# output varies from run to run once tile threads are used (tools/soak_streams.py found it on the hardware, profiles/r06_v29_stream_soak_reference_race.txt; it is
# deterministic single-threaded).  The glue leaves that one decision open until the picture is parsed (glue/m355_glue.cc RBF_ROTATE_PENDING), so with ANY number
# of threads the backend shows what the reference decodes single-threaded.  (w, h, bd, tc, tr, frames, seed, intra, b, sao, features, slices, geom: three of the
# soak's streams)
ROTATE_LOOKUP_CASES = [(256, 128, 8, 3, 1, 2, 594, 3, 0, 0, 290, 2, 56), (768, 256, 8, 2, 1, 4, 307, 3, 0, 1, 276, 1, 0), (640, 256, 8, 2, 1, 6, 70, 3, 0, 0, 8532, 1, 97)]


def check_rotate_lookup(ref, tmp_path, case, backend, runs):
    w, h, bd, tc, tr, frames, seed, intra, b, sao, feat, slices, geom = case
    data = make_stream(tmp_path, w, h, bd, tc, tr, frames, seed, intra, b, sao, feat, 2, slices, geom)
    want = de265_py.decode_stream(ref, data, threads=0, scalar=True)
    assert want[1] == frames and not want[2]
    lib = glue_lib()
    for threads in [8] * runs + [0, 3]:
        got = de265_py.decode_stream(lib, data, threads=threads)
        assert got[:2] == want[:2], "threads %d: the backend differs from the single-threaded reference" % threads
    assert lib.m355_glue_cpu_pixel_calls() == 0
    assert os.path.realpath(lib.m355_glue_backend_path().decode()) == os.path.realpath(backend)


@pytest.mark.parametrize("case", ROTATE_LOOKUP_CASES[:1], ids=lambda c: "seed%d" % c[6])
def test_chroma_rotation_lookup_is_thread_independent_emulated_backend(ref, emu_lib, tmp_path, monkeypatch, case):  # noqa: F811
    monkeypatch.setenv("M355_LIB", EMU_SO)
    check_rotate_lookup(ref, tmp_path, case, EMU_SO, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ROTATE_LOOKUP_CASES, ids=lambda c: "seed%d" % c[6])
def test_chroma_rotation_lookup_is_thread_independent_gpu(ref, tmp_path, monkeypatch, case):
    monkeypatch.delenv("M355_LIB", raising=False)
    check_rotate_lookup(ref, tmp_path, case, capi.DEFAULT_LIB, 8)
