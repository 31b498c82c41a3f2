"""Parity cases with k_intra's dependency levels derived from what each intra MODE can read (runtime_upload.hip intra_schedule): python one_sided_worker.py <library .so or "default"> <oracle .so>.  Exit code 0 = every picture
equals the oracle's.  A dependency dropped wrongly lets a block run before (or beside) a block it reads from: the interpreter's
shuffled wave order and non-zero memory turn that into different samples."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from libde265_amd import capi  # noqa: E402
from oracle_py import Oracle  # noqa: E402
from synth_util import assert_planes_equal, device_decode, make_case, oracle_decode  # noqa: E402

# all-intra pictures (every mode, every size, 4x4 .. 32x32, NxN) over CTB sizes, chroma formats (4:4:4 smooths chroma borders), bit
# depths, tiles / slices (unavailable neighbours -> substitution), PCM, constrained intra prediction (the switch must stand back),
# and sparse intra blocks among inter ones
CASES = [dict(width=256, height=192, bit_depth=8, seed=1201, intra_pct=100, n_refs=0),
         dict(width=192, height=128, bit_depth=10, seed=1202, intra_pct=100, n_refs=0, log2_ctb=5),
         dict(width=136, height=104, bit_depth=8, seed=1203, intra_pct=100, n_refs=0, log2_ctb=4),
         dict(width=256, height=128, bit_depth=8, seed=1204, intra_pct=100, n_refs=0, tile_cols=2, tile_rows=2, n_slices=3),
         dict(width=192, height=128, bit_depth=8, seed=1205, intra_pct=100, n_refs=0, chroma_format=3),
         dict(width=192, height=128, bit_depth=10, seed=1206, intra_pct=100, n_refs=0, chroma_format=2),
         dict(width=192, height=128, bit_depth=8, seed=1207, intra_pct=100, n_refs=0, features=8 + 16),
         dict(width=192, height=128, bit_depth=8, seed=1208, intra_pct=60, features=1),
         dict(width=256, height=192, bit_depth=8, seed=1209, intra_pct=25),
         dict(width=128, height=128, bit_depth=12, seed=1210, intra_pct=100, n_refs=0, fixed_cu_log2=3),
         dict(width=128, height=128, bit_depth=8, seed=1211, intra_pct=100, n_refs=0, fixed_cu_log2=5)]
BIG = [dict(width=1920, height=1080, bit_depth=8, seed=1220, intra_pct=100, n_refs=0), dict(width=3840, height=2160, bit_depth=10, seed=1221, tile_cols=4, tile_rows=2, intra_pct=5)]


if __name__ == "__main__":
    big = sys.argv[1] == "default"
    lib = capi.Library() if big else capi.Library(sys.argv[1])
    o = Oracle(ctypes.CDLL(sys.argv[2]))
    for case in CASES + (BIG if big else []):
        pic, refs = make_case(**case)
        want = oracle_decode(o, pic, refs)
        ctx = capi.Context(lib, 0)
        try:
            # (an idle pipeline at any depth: k_intra with its halo keeper wave, planning an intra picture's CTBs itself; M355_TEST_NO_KEEPER=1 or pictures
            # in flight: the 12-wave kernel behind the planner's launch)
            for depth in (1, 3):
                ctx.set_pipeline_depth(depth)
                assert_planes_equal(device_decode(ctx, pic, refs), want, "seed %d depth %d" % (case["seed"], depth))
                assert_planes_equal(device_decode(ctx, pic, refs, resident=True, repeat=2), want, "seed %d depth %d resident" % (case["seed"], depth))
        finally:
            ctx.close()
    print("one-sided worker ok")
