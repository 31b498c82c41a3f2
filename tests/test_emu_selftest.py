"""The SIMT interpreter checked against itself (tests/simt_emu/selftest.cc): kernels with a missing barrier between waves, with a
dependency on block order, and a read of unwritten memory must FAIL under it; their correct twins must pass.  This is what gives the
emulated parity tests of the CPU tier their meaning beyond "the arithmetic is right"."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
EMU = os.path.join(HERE, "simt_emu")


def test_interpreter_catches_missing_barriers_block_order_and_unwritten_memory(tmp_path):
    exe = os.path.join(str(tmp_path), "selftest")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-x", "c++", "-I", EMU, os.path.join(EMU, "selftest.cc"), os.path.join(EMU, "simt_emu.cpp"), "-o", exe],
                   check=True)
    env = {k: v for k, v in os.environ.items() if not k.startswith("SIMT_EMU_")}
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    # ... and with the permutations and the fill switched off the wrong kernels get away with it, as they did before
    r = subprocess.run([exe], env=dict(env, SIMT_EMU_ORDER="linear", SIMT_EMU_WAVE_ORDER="linear", SIMT_EMU_POISON="0"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and r.stdout.count("FAIL:") == 3, r.stdout + r.stderr
