"""The horizontal-edge deblocking pass inside the SAO kernel (k_sao.hip d_dbh_block / k_sao_dbh; runtime.hip decode_post,
M355_FUSE_DBH=1) — an EXPERIMENTAL switch, off by default, written after the round's GPU minutes were spent: bit-exact against the
oracle under the SIMT interpreter; the GPU case is opt-in until tools/gpu_r5b.sh has taken it to hardware.  The switch is read once
per process: tests/dbh_worker.py runs in a process of its own."""
import os
import subprocess
import sys

import pytest

from test_emu_picture import EMU_SO, emu_lib  # noqa: F401  (fixture)

HERE = os.path.dirname(os.path.abspath(__file__))


def test_fused_horizontal_deblock_and_sao_emulated(emu_lib, oracle):  # noqa: F811
    r = subprocess.run([sys.executable, os.path.join(HERE, "dbh_worker.py"), EMU_SO, oracle._name], env=dict(os.environ, M355_FUSE_DBH="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "M355_FUSE_DBH=1" in r.stdout


@pytest.mark.gpu
@pytest.mark.skipif(not os.environ.get("M355_TEST_FUSE_DBH"), reason="opt-in (M355_TEST_FUSE_DBH=1): the switch has not seen hardware yet — tools/gpu_r5b.sh is its first visit")
def test_fused_horizontal_deblock_and_sao_gpu(oracle):
    r = subprocess.run([sys.executable, os.path.join(HERE, "dbh_worker.py"), "default", oracle._name], env=dict(os.environ, M355_FUSE_DBH="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
