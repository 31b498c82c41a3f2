"""k_intra's samples from NEIGHBOUR CTBs that arrive after the CTB was staged: on the hardware a CTB's prologue runs long before its
neighbours finish, so their samples reach it later — through the halo keeper (the workgroup's last wave re-reads the missing granules once
per level and stores what has arrived into the halo, k_intra.hip) or through the reading block's own poll.  The SIMT interpreter runs
k_intra's workgroups one after the other (a neighbour's samples are always there at staging time): M355_TEST_HALO_LATE=1 (read once per
process) makes the prologue take NO neighbour sample, so every one comes through those two paths — a keeper slot mapped to the wrong
halo entry or granule shows as different samples.  The planner's pruning (border entries a mode never reads point at the constant cell,
k_common.h m355_intra_used_entries) is covered by every intra parity test; here it meets the late halo.  GPU: the same pictures + the
1080p / 4K ones, where the lateness is real."""
import os
import subprocess
import sys

import pytest

from test_emu_picture import EMU_SO, emu_lib  # noqa: F401  (fixture)

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("no_keeper", [False, True], ids=["keeper", "own_polls"])
def test_late_halo_samples_emulated(emu_lib, oracle, no_keeper):  # noqa: F811
    # own_polls: M355_TEST_NO_KEEPER=1 — the 12-wave kernel behind the planner's launch (what a picture gets when others are in flight: the
    # interpreter's pipeline is never busy), where every late sample is fetched by the block that reads it
    env = dict(os.environ, M355_TEST_HALO_LATE="1")
    if no_keeper: env["M355_TEST_NO_KEEPER"] = "1"
    r = subprocess.run([sys.executable, os.path.join(HERE, "one_sided_worker.py"), EMU_SO, oracle._name], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "one-sided worker ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
@pytest.mark.parametrize("no_keeper", [False, True], ids=["keeper", "own_polls"])
def test_late_halo_samples_gpu(oracle, no_keeper):
    env = dict(os.environ, M355_TEST_HALO_LATE="1")
    if no_keeper: env["M355_TEST_NO_KEEPER"] = "1"
    r = subprocess.run([sys.executable, os.path.join(HERE, "one_sided_worker.py"), "default", oracle._name], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "one-sided worker ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
