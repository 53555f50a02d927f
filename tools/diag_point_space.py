"""Forward error of the first set-abstraction layer against float64: gather GEMM vs point space."""
import torch
from scan2cap_amd.pointnet2 import fused

torch.manual_seed(0)
for (B, n, m, ns, C, N) in ((2, 8192, 2048, 64, 132, 64), (2, 2048, 1024, 32, 128, 128)):
    xyz = torch.rand(B, n, 3, device="cuda") * 6 - 3
    new_xyz = xyz[:, :m].contiguous()
    cloud = torch.randn(B, n, 3 + C, device="cuda")
    feats = cloud[..., 3:]
    idx = torch.randint(0, n, (B, m, ns), device="cuda", dtype=torch.int32)
    W = torch.randn(N, 3 + C, device="cuda") / (3 + C) ** 0.5
    g = fused.GatherSpec(xyz, new_xyz, feats, idx, 0.2, True)
    M = g.rows
    Y0 = torch.empty(M, N, device="cuda")
    fused._call("s2c_sa_gather_gemm", Y0, g.B, g.N, g.m, g.ns, g.C, g.frs, g.fbs, g.radius, g.normalize,
                g.xyz.data_ptr(), g.new_xyz.data_ptr(), g.feats.data_ptr(), g.idx.data_ptr(), N,
                W.data_ptr(), W.stride(0), Y0.data_ptr(), N, None)
    f2 = g.feats2d()
    P = torch.empty(B * n, N, device="cuda")
    fused._call("s2c_rows_gemm", P, B * n, N, C, f2.data_ptr(), f2.stride(0), W[:, 3:].data_ptr(),
                W.stride(0), None, None, P.data_ptr(), N, None)
    Y1 = torch.empty(M, N, device="cuda")
    fused._call("s2c_sa_gather_add", Y1, B, n, m, ns, N, g.radius, g.normalize, g.xyz.data_ptr(),
                g.new_xyz.data_ptr(), P.data_ptr(), idx.data_ptr(), W.data_ptr(), W.stride(0),
                Y1.data_ptr(), None)
    X = g.materialise().double()
    want = X @ W.double().t()
    P64 = f2.double() @ W[:, 3:].double().t()
    Yt = (X.float() @ W.t())
    for name, Y in (("gather gemm", Y0), ("point space", Y1), ("torch fp32 mm", Yt)):
        e = (Y.double() - want)
        print("%-14s max %.3e rms %.3e  (|Y| rms %.3f)" % (name, e.abs().max(), e.pow(2).mean().sqrt(),
                                                          want.pow(2).mean().sqrt()))
    print("old vs new:   %.3f of the values differ, rms %.3e" % (float((Y0 != Y1).float().mean()),
          float((Y0 - Y1).double().pow(2).mean().sqrt())))
    for name, Y in (("gather gemm", Y0), ("point space", Y1)):
        e = (Y.double() - want)
        cm = e.mean(0)            # per-channel mean signed error
        print("%-14s per-channel bias: rms %.3e max %.3e (column rms error %.3e)" % (
            name, cm.pow(2).mean().sqrt(), cm.abs().max(), e.pow(2).mean().sqrt()))
    e = P.double() - P64
    print("P              max %.3e rms %.3e" % (e.abs().max(), e.pow(2).mean().sqrt()))
