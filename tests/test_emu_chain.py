"""CPU tier of tests/test_gpu_pipeline.py::test_long_unsynchronised_chain: the bookkeeping of the completion marks (one per decode,
a ring of 256 that wraps here: 340 marks) under the SIMT interpreter — every launch is synchronous there, so this checks the ring's takeover
and the handle / frame recycling, not the ordering (that is the GPU tier's)."""
from test_emu_picture import emu_lib  # noqa: F401  (fixture)
from test_gpu_pipeline import long_chain


def test_long_chain_emulated(emu_lib, oracle):  # noqa: F811
    long_chain(emu_lib, oracle, dict(width=128, height=64, bit_depth=8, seed=212, n_refs=2), 3, 170)   # (two marks per step: the list copy and the decode)
