import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scan2cap_amd.pointnet2 import fused
J, ns, K, C3 = 2048, 64, 64, 128
torch.manual_seed(0)
M = J * ns
A = torch.relu(torch.randn(M, K, device="cuda"))
arg = torch.randint(0, ns, (J, C3), device="cuda", dtype=torch.int32)
dk = torch.randn(J, C3, device="cuda")
W = torch.randn(C3, K, device="cuda") * 0.2
for name, G, cv in (("sparse only", torch.zeros(K, K, device="cuda"), torch.zeros(K, device="cuda")),
                    ("dense only", torch.randn(K, K, device="cuda") * 0.1, torch.randn(K, device="cuda"))):
    dkk = dk if name == "sparse only" else torch.zeros_like(dk)
    a16 = arg.to(torch.int16)
    Wcat = torch.cat([(-G.t()), W.t()], 1).contiguous()
    dA = torch.full((M, K), float("nan"), device="cuda")
    fused._call("s2c_pool_bwd_input_grad", A, M, K, K, C3, ns, A.data_ptr(), K, a16.data_ptr(), dkk.data_ptr(),
                Wcat.data_ptr(), Wcat.stride(0), cv.data_ptr(), dA.data_ptr(), K, None, None, 0)
    torch.cuda.synchronize()
    dense = torch.zeros(M, C3, device="cuda", dtype=torch.float64)
    rows = (torch.arange(J, device="cuda").unsqueeze(1) * ns + arg.long())
    dense.view(-1)[(rows * C3 + torch.arange(C3, device="cuda")).view(-1)] = dkk.double().view(-1)
    ref = dense @ W.double() - A.double() @ G.double() + cv.double()
    err = (dA.double() - ref).abs().max(1)[0]
    bad = (err > 1e-4).nonzero().flatten()
    print(name, "max err", float(err.max()), "bad rows", bad.numel(), "first", bad[:10].tolist(), "bad%64", sorted(set((bad % 64).tolist()))[:70])
from tools.bench_ops import timeit
for (J, ns, K, C3) in ((16384, 64, 64, 128), (16384, 64, 64, 64), (16384, 64, 64, 16)):
    M = J * ns
    A = torch.relu(torch.randn(M, K, device="cuda"))
    arg = torch.randint(0, ns, (J, C3), device="cuda", dtype=torch.int32)
    dk = torch.randn(J, C3, device="cuda"); arg16 = arg.to(torch.int16)
    W = torch.randn(C3, K, device="cuda") * 0.2
    G = torch.randn(K, K, device="cuda") * 0.1; cv = torch.randn(K, device="cuda")
    Wcat = torch.cat([(-G.t()), W.t()], 1).contiguous()
    dA = torch.empty((M, K), device="cuda")
    f = lambda: fused._call("s2c_pool_bwd_input_grad", A, M, K, K, C3, ns, A.data_ptr(), K, arg16.data_ptr(), dk.data_ptr(),
                            Wcat.data_ptr(), Wcat.stride(0), cv.data_ptr(), dA.data_ptr(), K, None, None, 0)
    print("M=%d K=%d C3=%d: %.1f us" % (M, K, C3, timeit(f, iters=10)))
