import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# The suite runs in the LIBRARY's default arithmetic -- the reference-precision mode (bf16x3), the mode bench.py's headline is measured in --
# unless AOD_CONV_PREC says otherwise.  Every test starts in that mode (autouse fixture below: a test that switched modes cannot leak its
# choice into the next one); tests of the FAST mode's own kernels (bit-equality of the fused / grouped / unfused bf16 forms, bf16 tolerances)
# pin it with the `bf16_mode` fixture, tests that compare the two modes switch explicitly.  Child processes (drivers, multi-rank workers)
# inherit AOD_CONV_PREC.
DEFAULT_PREC = os.environ.setdefault('AOD_CONV_PREC', 'bf16x3')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # the RCCL test first: a lease with >= 2 visible devices meets the first-run collective code before anything else
    items.sort(key=lambda it: 0 if 'rccl' in it.name.lower() else 1)


def _set(p):
    AF = sys.modules.get('aod_meh_hua_amd.functional')
    if AF is None:
        if p == DEFAULT_PREC:
            return                      # (not imported yet: it will come up in the default mode)
        from aod_meh_hua_amd import functional as AF
    AF.set_precision(p)


@pytest.fixture(autouse=True)
def _default_precision():
    _set(DEFAULT_PREC)
    yield
    _set(DEFAULT_PREC)


@pytest.fixture
def bf16_mode(_default_precision):
    """the fast mode (plain bf16 operands) for tests of its own kernels"""
    _set('bf16')
    yield
    _set(DEFAULT_PREC)
