"""Pictures of 16 384 prediction blocks and more clear their metadata planes inside k_job_count and, on a single-stream lane, run the
CU plane, the SAO masks and k_inter_jobs' job list as roles of ONE launch (k_meta_planes_jobs, runtime_decode.hip launch_prediction).  The CPU
tier's pictures are smaller than that: M355_CLEAR_IN_COUNT_MIN=1 (read once per process) sends them down the same path under the SIMT
interpreter — the synthetic and the random picture suites once more, in a process of their own."""
import os
import subprocess
import sys

from test_emu_picture import emu_lib  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_merged_meta_launch_emulated(emu_lib):  # noqa: F811
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_emu_synth.py"),
                        os.path.join(ROOT, "tests", "test_emu_random.py"), "-k", "not seed1 and not seed3 and not seed5 and not seed7"],
                       env=dict(os.environ, M355_CLEAR_IN_COUNT_MIN="1"), capture_output=True, text=True, timeout=1800, cwd=ROOT)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
