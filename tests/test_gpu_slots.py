"""GPU tier: every decoder slot of the table filled by init_acceleration_functions_mi355x() — the
reference's plugin interface for this path (acceleration.h:29-231, decctx.cc:239-270) — bit-exact
against the oracle on the reference's own test scenarios, 8/9/10/12 bit; plus the batched entry."""
import pytest

import slot_checks
from libde265_amd import capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    l = capi.Library()
    assert l.device_count() >= 1
    return l


@pytest.fixture(scope="module")
def table(lib):
    return capi.acceleration_functions(lib)


@pytest.mark.parametrize("check", slot_checks.ALL, ids=lambda f: f.__name__)
def test_slot_family(table, oracle, check):
    check(table, oracle, quick=False)


def test_transform_add_batch(lib, oracle):
    slot_checks.check_batch(lib, oracle, quick=False)


def test_slots_from_many_threads(table, oracle):
    """slots are called concurrently from the reference's worker pool (threads.h:86): per-thread staging"""
    import threading
    errs = []

    def work():
        try:
            slot_checks.check_deblock(table, oracle, quick=True)
            slot_checks.check_dequant(table, oracle, quick=True)
        except Exception as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
