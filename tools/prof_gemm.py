"""Phase timeline of one workgroup of the bf16x3 rows GEMM (s_memtime stamps written by
wave 0, csrc/s2c_gemm.hip: X3_STAMP) for the SA layer shapes."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd import _C
from scan2cap_amd.pointnet2 import fused  # noqa: F401  (registers signatures)
from tools.bench_ops import timeit

_I, _L, _P, _F = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_float
_C.register("s2c_rows_gemm", [_L, _I, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P])
_C.register("s2c_sa_gather_gemm", [_I, _I, _I, _I, _I, _L, _L, _F, _I, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P])
lib = _C.load()
lib.s2c_gemm_set_profile.argtypes = [_P, _I]
lib.s2c_rows_gemm_blocks.argtypes = [_L, _I]; lib.s2c_rows_gemm_blocks.restype = _I
TICK_US = float(os.environ.get("S2C_TICK_US", "0.01"))    # s_memtime: 100 MHz


def timeline(f, nblocks, label, slices):
    f(); torch.cuda.synchronize()
    t = timeit(f, iters=20)
    print("%s: %.1f us, %d row blocks" % (label, t, nblocks))
    prof = torch.zeros(64, dtype=torch.int64, device="cuda")
    for blk in (nblocks // 7, nblocks // 2, nblocks - 40):
        prof.zero_()
        lib.s2c_gemm_set_profile(prof.data_ptr(), blk)
        f(); torch.cuda.synchronize()
        lib.s2c_gemm_set_profile(None, 0)
        p = prof.cpu().numpy()
        n = int(p[63])
        d = (p[1:n] - p[:n - 1]) * TICK_US
        names = ["setup", "issue2"]
        for s in range(slices):
            names += ["s%d:bar" % s, "s%d:wait+split" % s, "s%d:bar" % s, "s%d:issue" % s, "s%d:mfma" % s]
        names += ["epilogue", "stores-acked"]
        print("  block %6d  life %.2f us: " % (blk, (p[n - 1] - p[0]) * TICK_US) +
              "  ".join("%s %.2f" % (nm, x) for nm, x in zip(names, d)))


def plain(M, N, K):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1
    Y = torch.empty(M, N, device="cuda")
    nb = lib.s2c_rows_gemm_blocks(M, N)
    part = torch.empty(nb * 2 * N, device="cuda")
    f = lambda: _C.call("s2c_rows_gemm", M, N, K, A.data_ptr(), K, W.data_ptr(), K, None, None,
                        Y.data_ptr(), N, part.data_ptr(), _C.stream_ptr())
    timeline(f, nb, "plain M=%d N=%d K=%d" % (M, N, K), (K + 31) // 32)


def gather(B, n, m, ns, C, N, local):
    pc = torch.randn(B, n, 3 + C, device="cuda")
    xyz = pc[..., :3].contiguous(); feats = pc[..., 3:]
    inds = torch.stack([torch.randperm(n, device="cuda")[:m] for _ in range(B)])
    new_xyz = torch.gather(xyz, 1, inds.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    if local:   # neighbours of centre j = a contiguous run of points
        idx = ((torch.arange(m, device="cuda").view(1, m, 1) * (n // m) +
                torch.arange(ns, device="cuda").view(1, 1, ns)) % n).expand(B, m, ns).to(torch.int32).contiguous()
    else:
        idx = torch.randint(0, n, (B, m, ns), device="cuda", dtype=torch.int32)
    W = torch.randn(N, 3 + C, device="cuda") * 0.1
    M = B * m * ns
    Y = torch.empty(M, N, device="cuda")
    nb = lib.s2c_rows_gemm_blocks(M, N); part = torch.empty(nb * 2 * N, device="cuda")
    f = lambda: _C.call("s2c_sa_gather_gemm", B, n, m, ns, C, feats.stride(1), feats.stride(0), 0.2, 1,
                        xyz.data_ptr(), new_xyz.data_ptr(), feats.data_ptr(), idx.data_ptr(), N,
                        W.data_ptr(), 3 + C, Y.data_ptr(), N, part.data_ptr(), _C.stream_ptr())
    timeline(f, nb, "gather B=%d n=%d m=%d ns=%d C=%d N=%d %s" % (B, n, m, ns, C, N, "local" if local else "random"),
             (3 + C + 31) // 32)


plain(1048576, 64, 64)
plain(1048576, 64, 128)
gather(8, 40000, 2048, 64, 132, 64, False)
gather(8, 40000, 2048, 64, 132, 64, True)
gather(8, 40000, 2048, 64, 61, 64, False)
plain(262144, 128, 128)
plain(262144, 256, 128)
