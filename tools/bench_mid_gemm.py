"""The mid-size layer GEMMs of the cfg3 step (2048 .. 65536 rows: FP modules, vote / proposal
heads, SA3 / SA4), each timed alone in a captured hipGraph of 20 calls: s2c_rows_gemm (with
statistics partials), torch.mm, and the algorithmic floors (HBM bytes at 5 TB/s, bf16x3 MFMA
flops at 2.5 PFLOP/s)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scan2cap_amd import _C
from scan2cap_amd.pointnet2 import fused  # noqa: F401

lib = _C.load()
lib.s2c_rows_gemm_blocks.argtypes = [ctypes.c_longlong, ctypes.c_int]
lib.s2c_rows_gemm_blocks.restype = ctypes.c_int
dev = torch.device("cuda:0")
SHAPES = [(32768, 128, 128), (65536, 128, 128), (65536, 256, 128), (32768, 256, 128),
          (8192, 256, 256), (8192, 256, 512), (4096, 256, 512), (4096, 256, 256),
          (2048, 128, 128), (20480, 128, 256), (20480, 128, 128), (20480, 256, 128),
          (8192, 259, 256), (8192, 256, 259), (65536, 128, 259), (32768, 128, 259)]
if len(sys.argv) > 1:
    SHAPES = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best

tot = [0.0, 0.0, 0.0]
for M, N, K in SHAPES:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.1
    Y = torch.empty(M, N, device=dev); Y2 = torch.empty(M, N, device=dev)
    nb = lib.s2c_rows_gemm_blocks(M, N)
    part = torch.empty(nb * 2 * N, device=dev)
    def hand(p=part):
        _C.call("s2c_rows_gemm", M, N, K, A.data_ptr(), K, W.data_ptr(), K, None, None,
                Y.data_ptr(), N, p.data_ptr() if p is not None else None, _C.stream_ptr())
    t1 = timed(hand)
    t2 = timed(lambda: torch.mm(A, W.t(), out=Y2))
    ref = A.double() @ W.double().t()
    err = float((Y.double() - ref).abs().max() / ref.abs().max())
    hbm = 4.0 * (M * K + M * N + N * K) / 5e12 * 1e6
    mfma = 12.0 * M * N * K / 2.5e15 * 1e6
    tot[0] += t1; tot[1] += t2; tot[2] += max(hbm, mfma)
    print("(%6d,%4d,%4d)  hand %6.1f us   torch.mm %6.1f us   floors: hbm %5.1f mfma %5.1f   err %.1e"
          % (M, N, K, t1, t2, hbm, mfma, err))
print("total: hand %.1f us, library %.1f us, floor %.1f us" % tuple(tot))
