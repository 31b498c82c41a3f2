"""k_intra's work list made on the device (runtime.hip upload(), M355_DEVICE_WORKLIST; k_intra.hip k_work_keys / k_work_items) — an
EXPERIMENTAL switch, off by default: mode 2 = device and host lists compared inside the library for every upload, mode 1 = decode from
the device-made list alone.  The switch is read once per process: each mode runs tests/worklist_worker.py in a process of its own."""
import os
import subprocess
import sys

import pytest

from test_emu_picture import EMU_SO, emu_lib  # noqa: F401  (fixture)

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("mode", ["2", "1"])
def test_device_work_list_emulated(emu_lib, oracle, mode):  # noqa: F811
    r = subprocess.run([sys.executable, os.path.join(HERE, "worklist_worker.py"), EMU_SO, oracle._name], env=dict(os.environ, M355_DEVICE_WORKLIST=mode),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.skipif(not os.environ.get("M355_TEST_DEVICE_WORKLIST"), reason="opt-in (M355_TEST_DEVICE_WORKLIST=1): the switch has not seen hardware yet — tools/gpu_r5a.sh is its first visit")
@pytest.mark.parametrize("mode", ["2", "1"])
def test_device_work_list_gpu(oracle, mode):
    r = subprocess.run([sys.executable, os.path.join(HERE, "worklist_worker.py"), "default", oracle._name], env=dict(os.environ, M355_DEVICE_WORKLIST=mode),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]


def test_transform_edges_and_border_plans_in_one_launch_emulated(emu_lib, oracle):  # noqa: F811
    """M355_MERGE_TU_PLAN=1 (experiment, off by default): k_meta_tu's scatter and k_intra_plan's plans as one launch (k_tu_plan) — with
    M355_CLEAR_IN_COUNT_MIN=1, which sends the CPU tier's small pictures down the launch order of pictures of 16 384 prediction blocks and
    more (metadata-plane fill and k_intra's ticket reset inside k_job_count, launch_prediction in runtime.hip)"""
    r = subprocess.run([sys.executable, os.path.join(HERE, "worklist_worker.py"), EMU_SO, oracle._name],
                       env=dict(os.environ, M355_MERGE_TU_PLAN="1", M355_CLEAR_IN_COUNT_MIN="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
