"""rows_gemm (hand-written fp32 MFMA) vs torch.mm on the SA layer shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd import _C
from scan2cap_amd.pointnet2 import fused  # registers signatures
from tools.bench_ops import timeit

_I, _L, _P = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p
_C.register("s2c_rows_gemm", [_L, _I, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P])
lib = _C.load()
lib.s2c_rows_gemm_blocks.argtypes = [_L, _I]; lib.s2c_rows_gemm_blocks.restype = _I

def run(M, N, K, stats=True, pro=False):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1
    Y = torch.empty(M, N, device="cuda")
    nb = lib.s2c_rows_gemm_blocks(M, N)
    part = torch.empty(nb * 2 * N, device="cuda") if stats else None
    sc = torch.rand(K, device="cuda") + 0.5 if pro else None
    sh = torch.randn(K, device="cuda") * 0.1 if pro else None
    def f():
        _C.call("s2c_rows_gemm", M, N, K, A.data_ptr(), K, W.data_ptr(), K,
                sc.data_ptr() if pro else None, sh.data_ptr() if pro else None,
                Y.data_ptr(), N, part.data_ptr() if stats else None, _C.stream_ptr())
    f(); torch.cuda.synchronize()
    Ain = torch.relu(A * sc + sh) if pro else A
    ref = Ain.double() @ W.double().t()
    err = (Y.double() - ref).abs().max().item() / ref.abs().max().item()
    if stats:
        p = part.view(nb, 2, N).double().sum(0)
        e1 = (p[0] - ref.sum(0)).abs().max().item() / ref.sum(0).abs().max().item()
        e2 = (p[1] - (ref * ref).sum(0)).abs().max().item() / (ref * ref).sum(0).abs().max().item()
    else:
        e1 = e2 = 0
    t1 = timeit(f, iters=20)
    t2 = timeit(lambda: torch.mm(Ain, W.t()), iters=20)
    gf = 2.0 * M * N * K / 1e9
    byt = 4.0 * (M * K + M * N) / 1e9
    print(f"M={M:8d} N={N:4d} K={K:4d} pro={int(pro)}: mine {t1:8.1f} us ({gf/t1*1e3:6.1f} TF, {byt/t1*1e3:5.2f} TB/s)  torch.mm {t2:8.1f} us  relerr {err:.1e} stats {e1:.1e} {e2:.1e}")

DECODE = [(8192, 1536, 512), (8192, 1536, 300), (8192, 3500, 512), (8192, 300, 940),
          (2048, 1536, 512), (2048, 3500, 512)]
if os.environ.get("S2C_BENCH_DECODE"):
    for (M, N, K) in DECODE:
        run(M, N, K, stats=False)
    sys.exit(0)
for (M, N, K) in [(1048576, 64, 135), (1048576, 64, 64), (1048576, 128, 64), (262144, 128, 131),
                  (262144, 128, 128), (262144, 256, 128), (65536, 128, 259), (65536, 256, 128),
                  (32768, 128, 259), (8192, 256, 512), (1000, 64, 135)]:
    run(M, N, K)
_F = ctypes.c_float
_C.register("s2c_sa_gather_gemm", [_I, _I, _I, _I, _I, _L, _L, _F, _I, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P])

def run_gather(B, n, m, ns, C, N):
    pc = torch.randn(B, n, 3 + C, device="cuda")
    xyz = pc[..., :3].contiguous(); feats = pc[..., 3:]
    inds = torch.stack([torch.randperm(n, device="cuda")[:m] for _ in range(B)])
    new_xyz = torch.gather(xyz, 1, inds.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    idx = torch.randint(0, n, (B, m, ns), device="cuda", dtype=torch.int32)
    W = torch.randn(N, 3 + C, device="cuda") * 0.1
    M = B * m * ns
    Y = torch.empty(M, N, device="cuda")
    nb = lib.s2c_rows_gemm_blocks(M, N); part = torch.empty(nb * 2 * N, device="cuda")
    def f():
        _C.call("s2c_sa_gather_gemm", B, n, m, ns, C, feats.stride(1), feats.stride(0), 0.2, 1,
                xyz.data_ptr(), new_xyz.data_ptr(), feats.data_ptr(), idx.data_ptr(), N,
                W.data_ptr(), 3 + C, Y.data_ptr(), N, part.data_ptr(), _C.stream_ptr())
    f(); torch.cuda.synchronize()
    X = fused._GatherRows.apply(xyz, new_xyz, feats, idx, 0.2, True)
    ref = X.double() @ W.double().t()
    err = (Y.double() - ref).abs().max().item() / ref.abs().max().item()
    t1 = timeit(f, iters=20)
    t2 = timeit(lambda: torch.mm(fused._GatherRows.apply(xyz, new_xyz, feats, idx, 0.2, True), W.t()), iters=20)
    print(f"gather-gemm B={B} n={n} m={m} ns={ns} C={C} N={N}: fused {t1:8.1f} us  gather_rows+mm {t2:8.1f} us  relerr {err:.1e}")

run_gather(8, 40000, 2048, 64, 132, 64)
run_gather(8, 2048, 1024, 32, 128, 128)
run_gather(8, 1024, 512, 16, 256, 128)
run(1048576, 64, 64, pro=True)
run(262144, 256, 128, pro=True)
