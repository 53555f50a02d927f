"""Achieved HBM bandwidth of the point-major SA kernels at the SA1 / SA2 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd import _C
from scan2cap_amd.pointnet2 import fused
from scan2cap_amd.pointnet2.fused import _call, _ptr, _stat_blocks
from tools.bench_ops import timeit

def report(name, us, nbytes):
    print("%-28s %9.1f us  %6.2f TB/s (%.0f MB)" % (name, us, nbytes / us / 1e6, nbytes / 1e6))

for (J, ns, C) in [(16384, 64, 128), (16384, 64, 64), (8192, 32, 256)]:
    M = J * ns
    print("--- J=%d ns=%d C=%d (M=%d)" % (J, ns, C, M))
    Y = torch.randn(M, C, device="cuda"); dA = torch.randn(M, C, device="cuda")
    A = torch.empty_like(Y); dY = torch.empty_like(Y)
    sc = torch.rand(C, device="cuda") + .5; sh = torch.randn(C, device="cuda") * .1
    mean = torch.randn(C, device="cuda") * .1; inv = torch.rand(C, device="cuda") + .5
    gamma = torch.rand(C, device="cuda") + .5
    nb = _stat_blocks(M)
    part = torch.empty(nb * 2 * C, device="cuda"); coef = torch.empty(3 * C, device="cuda")
    dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
    rm = torch.zeros(C, device="cuda"); rv = torch.ones(C, device="cuda")
    t = timeit(lambda: _call("s2c_bn_train_stats", Y, M, C, Y.data_ptr(), part.data_ptr(), 1e-5, 0.1, gamma.data_ptr(), sh.data_ptr(), rm.data_ptr(), rv.data_ptr(), sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), None))
    report("bn_train_stats", t, 4 * M * C)
    t = timeit(lambda: _call("s2c_bn_relu", Y, M, C, Y.data_ptr(), sc.data_ptr(), sh.data_ptr(), A.data_ptr(), 1))
    report("bn_relu", t, 8 * M * C)
    t = timeit(lambda: _call("s2c_bn_relu_bwd", Y, M, C, dA.data_ptr(), Y.data_ptr(), sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(), 1, 0, part.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr(), dY.data_ptr()))
    report("bn_relu_bwd (stats+apply)", t, 4 * 5 * M * C)
    ym = torch.empty(J, C, device="cuda"); out = torch.empty(J, C, device="cuda"); arg = torch.empty(J, C, dtype=torch.int32, device="cuda")
    t = timeit(lambda: _call("s2c_bn_relu_max", Y, J, ns, C, Y.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(), arg.data_ptr(), ym.data_ptr()))
    report("bn_relu_max", t, 4 * M * C)
    dO = torch.randn(J, C, device="cuda")
    t = timeit(lambda: _call("s2c_bn_relu_max_bwd", Y, J, ns, C, dO.data_ptr(), arg.data_ptr(), ym.data_ptr(), Y.data_ptr(), sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(), 0, part.data_ptr(), coef.data_ptr(), dg.data_ptr(), db.data_ptr(), dY.data_ptr()))
    report("bn_relu_max_bwd", t, 4 * 2 * M * C)
