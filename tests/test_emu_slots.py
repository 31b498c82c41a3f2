"""CPU tier: the slot layer (init_acceleration_functions_mi355x, the reference's own plugin interface,
acceleration.h:29-231) with the product's slot kernels compiled unchanged under the SIMT interpreter,
against the oracle.  Reduced case lists (the interpreter is slow); the GPU tier runs the full lists."""
import pytest

import slot_checks
from test_emu_picture import emu_lib  # noqa: F401  (session fixture: builds the interpreter library)
from libde265_amd import capi


@pytest.fixture(scope="module")
def table(emu_lib):  # noqa: F811
    return capi.acceleration_functions(emu_lib)


@pytest.mark.parametrize("check", slot_checks.ALL, ids=lambda f: f.__name__)
def test_slot_family(table, oracle, check):
    check(table, oracle, quick=True)


def test_transform_add_batch(emu_lib, oracle):  # noqa: F811
    slot_checks.check_batch(emu_lib, oracle, quick=True)
