"""Phase timeline (s_memtime stamps of wave 0, shader clock) of one workgroup of the 64-k-chunk
GEMM (rows_gemm_c64_kernel): start | loads issued | per chunk: landed+staged+barrier, MFMAs
(+barrier) ... | stores issued | stores acknowledged."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd import _C
from scan2cap_amd.pointnet2 import fused  # noqa: F401
lib = _C.load()
_I, _L, _P = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p
lib.s2c_gemm_set_profile.argtypes = [_P, _I]
lib.s2c_rows_gemm_blocks.argtypes = [_L, _I]; lib.s2c_rows_gemm_blocks.restype = _I
GHZ = 2.4
SHAPES = [(2048, 128, 128), (32768, 128, 128), (8192, 256, 512), (262144, 256, 128)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for M, N, K in SHAPES:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.1
    Y = torch.empty(M, N, device="cuda")
    nb = lib.s2c_rows_gemm_blocks(M, N); part = torch.empty(nb * 2 * N, device="cuda")
    f = lambda: _C.call("s2c_rows_gemm", M, N, K, A.data_ptr(), K, W.data_ptr(), K, None, None,
                        Y.data_ptr(), N, part.data_ptr(), _C.stream_ptr())
    f(); torch.cuda.synchronize()
    prof = torch.zeros(64, dtype=torch.int64, device="cuda")
    for blk in (0, nb // 2, nb - 1):
        prof.zero_(); lib.s2c_gemm_set_profile(prof.data_ptr(), blk)
        f(); torch.cuda.synchronize()
        lib.s2c_gemm_set_profile(None, 0)
        p = prof.cpu().numpy(); n = int(p[63])
        d = (p[1:n] - p[:n - 1]) / GHZ / 1e3
        print("(%d,%d,%d) block %5d life %.2f us: %s" % (M, N, K, blk, (p[n - 1] - p[0]) / GHZ / 1e3,
              " ".join("%.2f" % x for x in d)))
