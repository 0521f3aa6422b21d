"""GPU: the active-learning drivers run end to end on a small synthetic pool (2 cycles: train -> score pool -> select), RetinaNet and
SSD configs, through the runner with HIP-graph replay, checkpoints and the selection files."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('script,extra', [('tools/train_RetinaNet.py', ['--synthetic-size', '128', '--samples-per-gpu', '4']),
                                          ('tools/train_SSD.py', []),
                                          # the same cycles in the reference-precision mode (X-layout activations end to end)
                                          ('tools/train_RetinaNet.py', ['--synthetic-size', '128', '--samples-per-gpu', '4', '--precision', 'bf16x3']),
                                          ('tools/train_SSD.py', ['--precision', 'bf16x3'])])
def test_al_driver_two_cycles(script, extra, tmp_path):
    wd = f'pytest_{os.path.basename(script)[:-3]}_{os.getpid()}'
    cmd = [sys.executable, os.path.join(ROOT, script), '--synthetic', '48', '--cycles', '2', '--work-dir', wd] + extra
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    out = os.path.join(ROOT, 'work_dirs', wd)
    xl0, xl1 = np.load(os.path.join(out, 'X_L_0.npy')), np.load(os.path.join(out, 'X_L_1.npy'))
    unc = np.load(os.path.join(out, 'Unc_1.npy'))
    assert len(xl1) > len(xl0) and set(xl0) <= set(xl1) and unc.shape == (48,) and np.isfinite(unc).all()
    import shutil
    shutil.rmtree(out, ignore_errors=True)
