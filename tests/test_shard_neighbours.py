"""`shard.neighbour_ranks` (who a rank exchanges tile-boundary halos with): symmetric, never the rank itself, and it contains every
rank that owns a CTB within one CTB (edge or corner) of an own CTB — checked against the CTB owner map for a sweep of tile grids."""
import ctypes

import numpy as np
import pytest

from libde265_amd import capi, shard, worklist


def _pp(tc, tr, w_ctbs, h_ctbs):
    pp = np.zeros(1, worklist.PIC_PARAMS)
    pp["width"], pp["height"], pp["log2_ctb_size"] = 64 * w_ctbs, 64 * h_ctbs, 6
    pp["num_tile_cols"], pp["num_tile_rows"] = tc, tr
    for i in range(tc + 1):
        pp["col_bd"][0][i] = (i * w_ctbs) // tc
    for i in range(tr + 1):
        pp["row_bd"][0][i] = (i * h_ctbs) // tr
    return pp[0]


@pytest.mark.parametrize("tc,tr,nranks", [(4, 2, 8), (4, 2, 4), (4, 2, 3), (3, 1, 3), (1, 3, 2), (2, 2, 4), (5, 3, 6), (2, 1, 4), (6, 1, 3)])
def test_neighbour_ranks(tc, tr, nranks):
    pp = _pp(tc, tr, 12, 6)
    owner, w = shard.ctb_owner_map(pp, nranks)
    owner = owner.reshape(-1, w)
    want = [set() for _ in range(nranks)]
    h = owner.shape[0]
    for y in range(h):
        for x in range(w):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    if 0 <= y + dy < h and 0 <= x + dx < w and owner[y + dy, x + dx] != owner[y, x]:
                        want[owner[y, x]].add(int(owner[y + dy, x + dx]))
    for r in range(nranks):
        got = shard.neighbour_ranks(pp, r, nranks)
        assert r not in got and got == sorted(got)
        assert set(got) == want[r], (r, got, want[r])
        for q in got:
            assert r in shard.neighbour_ranks(pp, q, nranks)
        # the library's own version (m355_shard_peers: what m355_decode_sharded exchanges with) agrees
        lib = capi.Library()
        buf = (ctypes.c_int * 64)()
        raw = np.array([pp]).tobytes()
        n = lib.lib.m355_shard_peers(ctypes.c_char_p(raw), r, nranks, buf, 64)
        assert list(buf[:n]) == got
