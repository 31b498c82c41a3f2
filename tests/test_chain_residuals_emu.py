"""A dependent chain's picture transforms its residuals in its FRONT part — k_residual in front of k_inter, leaving int16 tiles — and adds them behind
k_inter (k_residual_add; runtime_decode.hip launch_prediction).  The runtime takes that order when a picture's reference is still being written, which the SIMT
interpreter never sees (it finishes every launch before the next call): M355_TEST_CHAIN_RESIDUALS=1 (read once per process) gives EVERY picture that order —
the synthetic, the random and the feature-stream picture suites once more, in a process of their own, alone and together with the merged metadata launches."""
import os
import subprocess
import sys

import pytest

from test_emu_picture import emu_lib  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("merged_meta", [False, True], ids=["plain", "merged_meta_launches"])
def test_front_part_residuals_emulated(emu_lib, merged_meta):  # noqa: F811
    env = dict(os.environ, M355_TEST_CHAIN_RESIDUALS="1")
    if merged_meta:
        env["M355_CLEAR_IN_COUNT_MIN"] = "1"
    # the synthetic suite — every feature, bit depth and chroma format — in both settings; the random pictures (half of the seeds) and the recorded
    # girlshy pictures once
    runs = [[os.path.join(ROOT, "tests", "test_emu_synth.py")]]
    if not merged_meta:
        runs.append([os.path.join(ROOT, "tests", "test_emu_random.py"), os.path.join(ROOT, "tests", "test_emu_picture.py"),
                     "-k", "not seed1 and not seed3 and not seed5 and not seed7 and not seed9"])
    for args in runs:
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider"] + args, env=env, capture_output=True, text=True, timeout=1800, cwd=ROOT)
        assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
