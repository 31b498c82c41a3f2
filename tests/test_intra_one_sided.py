"""k_intra's dependency levels come from what each intra MODE can read (runtime_upload.hip intra_schedule; the default since round 5: 42 % fewer
levels = barrier steps of the chain on the all-intra 1080p picture of BASELINE config 2, bit-exact on hardware, profiles/r05_a_*).
Bit-exact against the oracle under the SIMT interpreter (shuffled wave order, non-zero memory: a dependency dropped wrongly lets a block
run before or beside a block it reads from) and on the GPU, in a process of its own with the level statistics on: the picture with
constrained intra prediction must be scheduled the conservative way."""
import os
import subprocess
import sys

import pytest

from test_emu_picture import EMU_SO, emu_lib  # noqa: F401  (fixture)

HERE = os.path.dirname(os.path.abspath(__file__))


def test_mode_aware_intra_levels_emulated(emu_lib, oracle):  # noqa: F811
    r = subprocess.run([sys.executable, os.path.join(HERE, "one_sided_worker.py"), EMU_SO, oracle._name], env=dict(os.environ, M355_INTRA_LEVEL_STATS="1"),
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "(one-sided 1)" in r.stderr and "one-sided worker ok" in r.stdout
    # the picture with constrained intra prediction (seed 1208) is scheduled the old way
    assert "(one-sided 0)" in r.stderr


@pytest.mark.gpu
def test_mode_aware_intra_levels_gpu(oracle):
    r = subprocess.run([sys.executable, os.path.join(HERE, "one_sided_worker.py"), "default", oracle._name], env=dict(os.environ, M355_INTRA_LEVEL_STATS="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "(one-sided 1)" in r.stderr and "(one-sided 0)" in r.stderr
