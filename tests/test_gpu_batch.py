"""m355_decode_batch on the GPU: several independent intra pictures through ONE k_intra launch (k_intra<BATCH>) are, picture by
picture, the oracle's decode; batches back to back without host synchronisation on recycled lanes; ragged batches; refusals."""
import os
import subprocess
import sys

import pytest

from oracle_py import Oracle
from batch_util import check_batches, check_rejections
from libde265_amd import capi, worklist

pytestmark = pytest.mark.gpu

CASES = [
    (dict(width=416, height=240, bit_depth=8, seed=701), 2, [[0, 1], [1, 0], [0, 1]]),
    (dict(width=832, height=480, bit_depth=10, seed=702, tile_cols=2, tile_rows=2, features=31), 4, [[0, 1, 2, 3], [4, 5], [3, 2, 1, 0], [5]]),
    (dict(width=640, height=368, bit_depth=8, seed=703, chroma_format=3, log2_ctb=5), 3, [[0, 1, 2], [2, 0, 1]]),
    (dict(width=416, height=240, bit_depth=12, seed=704, chroma_format=2, n_slices=3, features=31), 8, [[0, 1, 2, 3, 4, 5, 6, 7], [7, 6, 5, 4, 3, 2, 1, 0]]),
    (dict(width=320, height=192, bit_depth=8, seed=705, chroma_format=4, sao=0), 2, [[0, 1], [1, 0]]),
    (dict(width=1920, height=1080, bit_depth=8, seed=706), 8, [[0, 1, 2, 3, 4, 5, 6, 7], [0, 1, 2, 3]]),
]


@pytest.mark.parametrize("cfg,depth,batches", CASES, ids=lambda v: "%dx%d_seed%d" % (v["width"], v["height"], v["seed"]) if isinstance(v, dict) else None)
def test_batched_intra_pictures_match_oracle(oracle, cfg, depth, batches):
    ctx = check_batches(capi.Library(), Oracle(oracle), cfg, depth, batches)[0]
    ctx.close()


def test_ragged_batch_and_prediction_only(oracle):
    cfg = dict(width=832, height=480, bit_depth=8, seed=711)
    ctx = check_batches(capi.Library(), Oracle(oracle), cfg, 3, [[0, 1, 2], [2, 1]], sizes={1: (64, 64), 2: (416, 960)})[0]
    ctx.close()
    st = worklist.STAGE_INTER | worklist.STAGE_RESIDUAL | worklist.STAGE_INTRA
    ctx = check_batches(capi.Library(), Oracle(oracle), cfg, 2, [[0, 1], [1, 0]], stages=st)[0]
    ctx.close()


def test_batches_and_single_decodes_interleaved(oracle):
    """the same pictures decoded many times, batches and single decodes mixed, no wait in between (the benchmark's pattern)"""
    cfg = dict(width=832, height=480, bit_depth=10, seed=721)
    ctx, pics, handles, frames, want = check_batches(capi.Library(), Oracle(oracle), cfg, 4, [[0, 1, 2, 3]])
    try:
        for it in range(6):
            ctx.decode_batch(handles[:4] if it % 2 == 0 else handles[1:4])
            ctx.decode_resident(handles[it % 4])
        ctx.wait()
        from synth_util import assert_planes_equal
        for k in range(4):
            assert_planes_equal(ctx.frame_download(frames[k]), want[k], "picture %d replayed" % k)
    finally:
        ctx.close()


def test_batch_refusals(oracle):
    check_rejections(capi.Library(), Oracle(oracle), dict(width=416, height=240, bit_depth=8, seed=731))


@pytest.mark.parametrize("streams", ["0", "1", "4"])
def test_batch_stream_modes(oracle, streams):
    """M355_BATCH_STREAMS (read once per process): 0 = the pictures' own stages on their lanes' streams, k_intra shared; 1 / 4 = whole
    batches on that many streams of their own"""
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "batch_worker.py"), "default", oracle._name], env=dict(os.environ, M355_BATCH_STREAMS=streams),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
