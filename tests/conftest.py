import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# The library's default arithmetic is the reference-precision mode (bf16x3).  The bulk of this suite pins the FAST mode's kernels (bit-equality
# of fused / grouped / unfused forms, bf16 tolerances against the goldens); the reference-precision mode has its own tests, which switch
# with functional.set_precision('bf16x3') and switch back to 'bf16'.  Child processes (drivers, multi-rank workers) inherit the setting.
# (`AOD_CONV_PREC=bf16x3 python -m pytest tests -m gpu` runs the same suite with the reference-precision kernels under every test that does not
# choose a mode itself: the fast mode's tolerances hold a fortiori, the bit-equality tests compare the x3 forms with each other.)
os.environ.setdefault('AOD_CONV_PREC', 'bf16')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
