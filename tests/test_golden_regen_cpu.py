"""The recipe that pins the oracle must run in ONE command: `python tests/golden/make_golden.py` (no arguments)
regenerates every fixture from the reference (one interpreter per generator) and the result equals what is
committed, array for array.  Needs /root/reference (build container only); nothing here runs on the GPU box."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


@pytest.mark.skipif(not os.path.isdir("/root/reference/python/sglang"), reason="the reference checkout is not here")
def test_make_golden_without_arguments_reproduces_the_committed_fixtures(tmp_path):
    env = dict(os.environ, SEMIPD_GOLDEN_OUT=str(tmp_path))
    p = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:]
    committed = sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLDEN, "*.npz")))
    made = sorted(os.path.basename(f) for f in glob.glob(os.path.join(str(tmp_path), "*.npz")))
    assert made == committed, f"generated {made}, committed {committed}"
    for name in committed:
        a, b = np.load(os.path.join(GOLDEN, name), allow_pickle=False), np.load(os.path.join(str(tmp_path), name), allow_pickle=False)
        assert sorted(a.files) == sorted(b.files), name
        for k in a.files:
            assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, (name, k)
            assert np.array_equal(a[k], b[k], equal_nan=(a[k].dtype.kind == "f")), (name, k)
