"""The fixtures under tests/golden/ are the parity root: this test re-runs their generator (tools/make_golden.py, which imports the
reference from /root/reference) on a clean output directory with ONE command and compares every array of every file with the committed
fixture byte for byte — dtype, shape and contents.  It only runs where the reference checkout exists (the build container); the GPU box has
no /root/reference and skips it."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/rlkit"), reason="the reference checkout is not on this machine")
def test_generator_reproduces_every_committed_fixture(tmp_path):
    env = dict(os.environ, ILSX_GOLDEN_OUT=str(tmp_path), OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_golden.py")], env=env, cwd=str(tmp_path), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    committed = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    assert committed and sorted(os.path.basename(f) for f in committed) == sorted(os.listdir(tmp_path))
    for f in committed:
        a, b = np.load(f, allow_pickle=True), np.load(os.path.join(tmp_path, os.path.basename(f)), allow_pickle=True)
        assert sorted(a.files) == sorted(b.files), f
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and a[k].tobytes() == b[k].tobytes(), (f, k)
