"""Runs one m355_decode_batch parity case in a process of its own (the library reads M355_BATCH_STREAMS once per process):
python batch_worker.py <library .so or "default"> <oracle .so>.  Exit code 0 = every picture equals the oracle's."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from batch_util import check_batches  # noqa: E402
from libde265_amd import capi  # noqa: E402
from oracle_py import Oracle  # noqa: E402

if __name__ == "__main__":
    lib = capi.Library() if sys.argv[1] == "default" else capi.Library(sys.argv[1])
    o = Oracle(ctypes.CDLL(sys.argv[2]))
    big = sys.argv[1] == "default"
    cfg = dict(width=832 if big else 192, height=480 if big else 128, bit_depth=10, seed=811, tile_cols=2, features=31)
    ctx = check_batches(lib, o, cfg, 4, [[0, 1, 2, 3], [4, 5], [3, 2, 1, 0], [5]])[0]
    ctx.close()
    print("batch worker ok (M355_BATCH_STREAMS=%s)" % os.environ.get("M355_BATCH_STREAMS", "unset"))
