"""CPU tier of tests/test_gpu_pipeline.py::test_long_unsynchronised_chain: the bookkeeping of the completion marks (one per decode,
a ring of 256 that wraps here: 340 marks) under the SIMT interpreter — every launch is synchronous there, so this checks the ring's takeover
and the handle / frame recycling, not the ordering (that is the GPU tier's)."""
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from test_gpu_pipeline import long_chain


def test_long_chain_emulated(emu_lib, oracle):  # noqa: F811
    long_chain(emu_lib, oracle, dict(width=128, height=64, bit_depth=8, seed=212, n_refs=2), 3, 170)   # (two marks per step: the list copy and the decode)


def test_chain_bookkeeping_emulated(emu_lib, oracle):  # noqa: F811
    """The path a dependent chain's picture takes on the hardware (runtime_decode.hip: front part on a spare lane, change of stream at the reference wait, destination
    hazards behind it, residual tiles in front) — M355_TEST_CHAIN_LANES=1 makes every decoded reference count as still being written, in a process of its own: chains
    at depths 2 (the whole picture on its reference's lane), 3 and 5, with and without SAO, two destination frames going round."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys, ctypes; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from test_emu_picture import EMU_SO\n"
            "from test_gpu_pipeline import long_chain\n"
            "from libde265_amd import capi\n"
            "lib = capi.Library(EMU_SO); o = ctypes.CDLL(%r)\n"
            "for depth in (2, 3, 5):\n"
            "    for sao in (0, 1):\n"
            "        long_chain(lib, o, dict(width=128, height=64, bit_depth=8, seed=215 + sao, n_refs=2, sao=sao), depth, 40, n_pool=2)\n"
            "        long_chain(lib, o, dict(width=192, height=128, bit_depth=10, seed=217 + sao, n_refs=2, sao=sao, intra_pct=20), depth, 24)\n"
            "print('chain bookkeeping ok')\n") % (here, os.path.dirname(here), oracle._name)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, M355_TEST_CHAIN_LANES="1"), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "chain bookkeeping ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
