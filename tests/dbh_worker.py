"""Parity cases with the horizontal-edge deblocking pass run INSIDE the SAO kernel (M355_FUSE_DBH=1, read once per process; k_sao.hip
d_dbh_block / k_sao_dbh, runtime.hip decode_post): python dbh_worker.py <library .so or "default"> <oracle .so>.
Exit code 0 = every picture equals the oracle's (all stages, and with the stage masks that switch the fused launch off again)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from libde265_amd import capi, worklist  # noqa: E402
from oracle_py import Oracle  # noqa: E402
from synth_util import assert_planes_equal, device_decode, make_case, oracle_decode  # noqa: E402

# what the fused pass has to get right: edges on every second lane-row boundary of a tile shifted by four rows (heights that are
# 0 / 4 / 8 / 12 mod 16), its rim columns (widths that are not multiples of 64 or 256), CTB rows inside a tile (16x16 / 32x32 CTBs),
# slices and tiles with and without filtering across them, PCM / bypass samples, chroma formats, bit depths (packed and 16-bit paths)
CASES = [dict(width=416, height=240, bit_depth=8, seed=1101, intra_pct=30),
         dict(width=200, height=136, bit_depth=8, seed=1102, log2_ctb=4, intra_pct=25),
         dict(width=328, height=200, bit_depth=10, seed=1103, log2_ctb=5, intra_pct=40, n_slices=3),
         dict(width=264, height=72, bit_depth=8, seed=1104, tile_cols=2, tile_rows=2, lf_across_tiles=0, intra_pct=50),
         dict(width=320, height=128, bit_depth=8, seed=1105, n_slices=3, tile_cols=3, intra_pct=15, features=31),
         dict(width=192, height=128, bit_depth=10, seed=1106, intra_pct=100, n_refs=0, features=24),
         dict(width=208, height=120, bit_depth=8, seed=1107, chroma_format=2, intra_pct=40),
         dict(width=208, height=104, bit_depth=10, seed=1108, chroma_format=3, intra_pct=40, features=32),
         dict(width=136, height=88, bit_depth=12, seed=1109, intra_pct=30),
         dict(width=160, height=96, bit_depth=16, seed=1110, intra_pct=30),
         dict(width=64, height=8, bit_depth=8, seed=1111, log2_ctb=4, intra_pct=60),
         dict(width=8, height=64, bit_depth=8, seed=1112, log2_ctb=4, intra_pct=60)]
BIG = [dict(width=1920, height=1080, bit_depth=8, seed=1120, tile_cols=2, tile_rows=2, intra_pct=10), dict(width=3840, height=2160, bit_depth=10, seed=1121, tile_cols=4, tile_rows=2, intra_pct=5)]
S = worklist


def one(lib, o, case):
    pic, refs = make_case(**case)
    for stages in (S.STAGE_ALL, S.STAGE_ALL & ~S.STAGE_SAO, S.STAGE_ALL & ~S.STAGE_DEBLOCK):
        want = oracle_decode(o, pic, refs, stages)
        ctx = capi.Context(lib, 0)
        try:
            got = device_decode(ctx, pic, refs, stages)
            assert_planes_equal(got, want, "seed %d stages %#x" % (case["seed"], stages))
            if stages == S.STAGE_ALL:
                got = device_decode(ctx, pic, refs, stages, resident=True, repeat=2)
                assert_planes_equal(got, want, "seed %d resident" % case["seed"])
        finally:
            ctx.close()


if __name__ == "__main__":
    big = sys.argv[1] == "default"
    lib = capi.Library() if big else capi.Library(sys.argv[1])
    o = Oracle(ctypes.CDLL(sys.argv[2]))
    for case in CASES + (BIG if big else []):
        one(lib, o, case)
    print("dbh worker ok (M355_FUSE_DBH=%s)" % os.environ.get("M355_FUSE_DBH", "unset"))
