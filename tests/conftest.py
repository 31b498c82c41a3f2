"""pytest configuration: markers + shared library fixtures.

Tiers (see the task contract):
  -m "not gpu" : oracle vs reference/golden vectors, host logic, ABI/symbol checks, SIMT-emulated
                 kernel-logic checks.  Runs without a GPU.
  -m gpu       : parity tests proper — HIP kernels through the C ABI vs the oracle.
"""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libde265_ref.so")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REFERENCE_ROOT = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # GPU tier: the tile-sharding tests hand torch tensors to the library.  torch ships its own HIP runtime;
    # when it initialises AFTER the library's (system) runtime in the same process it finds no device, the
    # other order works — so bring torch's up first (bench.py does the same for N > 1).
    expr = config.getoption("-m") or ""
    if ("gpu" in expr and "not gpu" not in expr) or any("test_gpu" in str(a) for a in config.args):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


def _make(target):
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "-j8", target], check=True,
                   stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/hevc_oracle.c). Built on demand — test infrastructure only."""
    src = os.path.join(ORACLE_DIR, "hevc_oracle.c")
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        _make("oracle")
    return ctypes.CDLL(ORACLE_SO)


@pytest.fixture(scope="session")
def ref():
    """The REAL reference compiled from /root/reference (oracle/_ref). Skips where neither the
    prebuilt .so nor the reference tree exists (e.g. a box without oracle/_ref)."""
    if not os.path.exists(REF_SO):
        if not os.path.isdir(REFERENCE_ROOT):
            pytest.skip("reference build (oracle/_ref) not available here")
        _make("ref")
    return ctypes.CDLL(REF_SO)
