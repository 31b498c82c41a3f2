"""CPU tier of m355_decode_batch (k_intra<BATCH> through the SIMT-interpreter build): several intra pictures in one launch give the
oracle's planes of each picture; ragged batches, batches back to back on recycled lanes, and what the call refuses."""
import os
import subprocess
import sys

import pytest

from oracle_py import Oracle
from batch_util import check_batches, check_rejections
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from libde265_amd import worklist

CASES = [
    (dict(width=128, height=128, bit_depth=8, seed=601), 2, [[0, 1]]),
    (dict(width=192, height=128, bit_depth=10, seed=602, tile_cols=2, features=31), 3, [[0, 1, 2], [3, 4], [5]]),
    (dict(width=128, height=128, bit_depth=8, seed=603, chroma_format=3, log2_ctb=5), 4, [[0, 1, 2, 3], [1, 0]]),
    (dict(width=136, height=72, bit_depth=8, seed=604, n_slices=2, chroma_format=4), 2, [[1, 0], [0, 1]]),
]


@pytest.mark.parametrize("cfg,depth,batches", CASES, ids=lambda v: "seed%d" % v["seed"] if isinstance(v, dict) else None)
def test_batched_intra_pictures_match_oracle(emu_lib, oracle, cfg, depth, batches):  # noqa: F811
    ctx = check_batches(emu_lib, Oracle(oracle), cfg, depth, batches)[0]
    ctx.close()


def test_ragged_batch_and_prediction_only(emu_lib, oracle):  # noqa: F811
    """pictures of different sizes in one batch (the shorter lists run out first); prediction + residual only (no filter stage behind
    the shared launch: k_intra writes the destination frames themselves)"""
    cfg = dict(width=192, height=128, bit_depth=8, seed=611)
    ctx = check_batches(emu_lib, Oracle(oracle), cfg, 3, [[0, 1, 2]], sizes={1: (64, 64), 2: (128, 192)})[0]
    ctx.close()
    st = worklist.STAGE_INTER | worklist.STAGE_RESIDUAL | worklist.STAGE_INTRA
    ctx = check_batches(emu_lib, Oracle(oracle), cfg, 2, [[0, 1], [1, 0]], stages=st)[0]
    ctx.close()


def test_batch_refusals(emu_lib, oracle):  # noqa: F811
    check_rejections(emu_lib, Oracle(oracle), dict(width=128, height=64, bit_depth=8, seed=621))


@pytest.mark.parametrize("streams", ["0", "2"])
def test_batch_stream_modes(emu_lib, oracle, streams):  # noqa: F811
    """M355_BATCH_STREAMS=0 (every picture's own stages on its lane, only k_intra shared) and a fixed number of batch streams: the
    setting is read once per process, so each runs in a process of its own"""
    from test_emu_picture import EMU_SO
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "batch_worker.py"), EMU_SO, oracle._name], env=dict(os.environ, M355_BATCH_STREAMS=streams),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
