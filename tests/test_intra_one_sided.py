"""k_intra's dependency levels from what each intra mode can read (runtime.hip intra_schedule, M355_INTRA_ONE_SIDED=1) — an EXPERIMENTAL
switch, off by default, written after the round's GPU minutes were spent: 42 % fewer levels (barrier steps of the chain) on the all-intra
1080p picture of BASELINE config 2.  Bit-exact against the oracle under the SIMT interpreter (shuffled wave order, non-zero memory);
the GPU case is opt-in until tools/gpu_r5c.sh has taken it to hardware.  The switch is read once per process."""
import os
import subprocess
import sys

import pytest

from test_emu_picture import EMU_SO, emu_lib  # noqa: F401  (fixture)

HERE = os.path.dirname(os.path.abspath(__file__))


def test_mode_aware_intra_levels_emulated(emu_lib, oracle):  # noqa: F811
    r = subprocess.run([sys.executable, os.path.join(HERE, "one_sided_worker.py"), EMU_SO, oracle._name], env=dict(os.environ, M355_INTRA_ONE_SIDED="1", M355_INTRA_LEVEL_STATS="1"),
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "(one-sided 1)" in r.stderr and "M355_INTRA_ONE_SIDED=1" in r.stdout
    # the picture with constrained intra prediction (seed 1208) is scheduled the old way
    assert "(one-sided 0)" in r.stderr


@pytest.mark.gpu
@pytest.mark.skipif(not os.environ.get("M355_TEST_INTRA_ONE_SIDED"), reason="opt-in (M355_TEST_INTRA_ONE_SIDED=1): the switch has not seen hardware yet — tools/gpu_r5c.sh is its first visit")
def test_mode_aware_intra_levels_gpu(oracle):
    r = subprocess.run([sys.executable, os.path.join(HERE, "one_sided_worker.py"), "default", oracle._name], env=dict(os.environ, M355_INTRA_ONE_SIDED="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
