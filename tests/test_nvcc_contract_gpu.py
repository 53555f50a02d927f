"""The nvcc-contraction variants of the index-producing kernels (libs2c_hip_nvcc1.so /
libs2c_hip_nvcc2.so: csrc/s2c_common.h sq3, built by scan2cap_amd.build with
S2C_NVCC_CONTRACT=1|2) against the oracle in the same mode: every FPS / ball-query / three_nn
parity test of tests/test_ops_gpu.py, bit-exact, in a process of its own (the variant is chosen
when the library is loaded).  The canonical build (mode 0) is what everything else tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [1, 2])
def test_index_ops_bit_exact_in_contraction_mode(mode):
    from scan2cap_amd import build as s2c_build
    assert os.path.exists(s2c_build.lib_path(mode)), \
        "run __graft_entry__.build() (or S2C_NVCC_CONTRACT=%d python -m scan2cap_amd.build)" % mode
    env = dict(os.environ, S2C_NVCC_CONTRACT=str(mode))
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_ops_gpu.py"),
                          "-m", "gpu", "-q", "-x", "-k", "fps or ball_query or three_nn"],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-1000:]
    assert " passed" in res.stdout
