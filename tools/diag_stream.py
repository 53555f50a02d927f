"""Which part of the streaming gather GEMM is off: feature columns, xyz columns, rows."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd import _C
from scan2cap_amd.pointnet2 import fused
_I, _L, _P, _F = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_float
_C.register("s2c_sa_gather_gemm", [_I, _I, _I, _I, _I, _L, _L, _F, _I, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P])
lib = _C.load()
lib.s2c_rows_gemm_blocks.argtypes = [_L, _I]; lib.s2c_rows_gemm_blocks.restype = _I
B, n, m, ns, C, N = 8, 40000, 2048, 64, 132, 64
torch.manual_seed(0)
pc = torch.randn(B, n, 3 + C, device="cuda")
xyz = pc[..., :3].contiguous(); feats = pc[..., 3:]
inds = torch.stack([torch.randperm(n, device="cuda")[:m] for _ in range(B)])
new_xyz = torch.gather(xyz, 1, inds.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
idx = torch.randint(0, n, (B, m, ns), device="cuda", dtype=torch.int32)
M = B * m * ns
X = fused._GatherRows.apply(xyz, new_xyz, feats, idx, 0.2, True)
for name, mask in (("all", None), ("xyz only", slice(0, 3)), ("feats only", slice(3, None)), ("feat 0", slice(3, 4)),
                   ("feat 131", slice(134, 135)), ("feat 127", slice(130, 131)), ("feat 128", slice(131, 132))):
    W = torch.randn(N, 3 + C, device="cuda") * 0.1
    if mask is not None:
        W2 = torch.zeros_like(W); W2[:, mask] = W[:, mask]; W = W2
    Y = torch.empty(M, N, device="cuda")
    nb = lib.s2c_rows_gemm_blocks(M, N); part = torch.empty(nb * 2 * N, device="cuda")
    _C.call("s2c_sa_gather_gemm", B, n, m, ns, C, feats.stride(1), feats.stride(0), 0.2, 1,
            xyz.data_ptr(), new_xyz.data_ptr(), feats.data_ptr(), idx.data_ptr(), N,
            W.data_ptr(), 3 + C, Y.data_ptr(), N, part.data_ptr(), _C.stream_ptr())
    torch.cuda.synchronize()
    ref = X.double() @ W.double().t()
    err = (Y.double() - ref).abs()
    bad_rows = (err.max(1)[0] > 1e-3 * ref.abs().max()).nonzero().flatten()
    print("%-10s max err %.3e (ref scale %.3e)  bad rows %d of %d  first bad %s  bad row %% 32: %s" % (
        name, err.max().item(), ref.abs().max().item(), bad_rows.numel(), M, bad_rows[:6].tolist(),
        sorted(set((bad_rows[:2000] % 32).tolist()))[:40]))
    p = part.view(nb, 2, N).double().sum(0)
    print("           stats err %.2e %.2e" % ((p[0] - ref.sum(0)).abs().max().item() / max(1e-9, ref.sum(0).abs().max().item()),
                                              (p[1] - (ref * ref).sum(0)).abs().max().item() / max(1e-9, (ref * ref).sum(0).abs().max().item())))
