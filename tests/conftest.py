import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """Plain `pytest tests` on a GPU-less host: GPU tests are skipped, not failed."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (no GPU on this host)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU parity oracle (oracle/s2c_oracle.c via ctypes) -- checker only."""
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def ext():
    """The product op layer (HIP, through the C ABI)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from scan2cap_amd.pointnet2 import _ext
    return _ext


@pytest.fixture(autouse=True)
def _persistent_grids_back_to_default():
    """`pipeline.GeometrySlots` sizes the persistent grids of the streaming kernels beside its geometry
    stage; a test that forgets `close()` would leave them changed for every later test -- and another
    grid size is another summation order of the BatchNorm partial sums, enough to move a golden
    gradient by 2e-4 (found as an order-dependent failure of tests/test_capnet_golden.py).  Reset
    after every test."""
    yield
    import torch
    if not torch.cuda.is_available() or os.environ.get("S2C_GEMM_STREAM_GRID"):
        return
    import ctypes
    from scan2cap_amd import _C
    lib = _C.load()
    for fn in ("s2c_gemm_set_stream_grid", "s2c_weight_grad_stream_set_grid"):
        f = getattr(lib, fn)
        f.argtypes = [ctypes.c_int]
        f(240)
