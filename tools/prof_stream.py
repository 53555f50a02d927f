"""Where one wave of the streaming GEMM (csrc/s2c_gemm2.hip) spends its cycles: waiting for its
ring chunks / k-steps (LDS reads, split, MFMAs) / epilogue, summed over its tiles.
The counters are shader-clock cycles, printed as us at GHZ = 2.4; prologue + tile loop add up to
~0.71 of the launch time measured by HIP events on every shape, i.e. the chip runs these kernels
at ~1.7 GHz: read the per-tile figures as cycles / 2400, not as wall time."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd import _C
from scan2cap_amd.pointnet2 import fused  # noqa: F401
lib = _C.load()
_I, _L, _P = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p
lib.s2c_gemm_stream_set_profile.argtypes = [_P, _I]
lib.s2c_rows_gemm_blocks.argtypes = [_L, _I]; lib.s2c_rows_gemm_blocks.restype = _I
GHZ = 2.4
dev = "cuda"

def report(label, f):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); f(); e1.record(); torch.cuda.synchronize()
    prof = torch.zeros(8, dtype=torch.int64, device=dev)
    for blk in (3, 120):
        prof.zero_(); lib.s2c_gemm_stream_set_profile(prof.data_ptr(), blk)
        f(); torch.cuda.synchronize()
        lib.s2c_gemm_stream_set_profile(None, 0)
        w, k, e, n, life, pro = [int(x) for x in prof.cpu()[:6]]
        n = max(n, 1)
        print("%s: %.1f us | block %3d wave 0: prologue %.1f us, %d tiles in %.1f us; per tile: wait %.2f us, k-steps %.2f us, epilogue %.2f us"
              % (label, e0.elapsed_time(e1) * 1e3, blk, pro / GHZ / 1e3, n, life / GHZ / 1e3, w / n / GHZ / 1e3, k / n / GHZ / 1e3, e / n / GHZ / 1e3))

M = 1048576
A = torch.randn(M, 64, device=dev); W = torch.randn(128, 64, device=dev) * 0.1
nb = lib.s2c_rows_gemm_blocks(M, 128)
part = torch.empty(nb * 2 * 128, device=dev)
Y = torch.empty(M, 128, device=dev)
report("plain (1M,128,64)", lambda: _C.call("s2c_rows_gemm", M, 128, 64, A.data_ptr(), 64, W.data_ptr(), 64, None, None,
                                            Y.data_ptr(), 128, part.data_ptr(), _C.stream_ptr()))
J = M // 64
ext = torch.empty(J, 128, device=dev); aext = torch.empty(J, 128, dtype=torch.int32, device=dev)
_C.register("s2c_rows_gemm_pool_raw", [_L, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P])
report("pool_raw (1M,128,64)", lambda: _C.call("s2c_rows_gemm_pool_raw", M, 128, 64, A.data_ptr(), 64, None, None, 0, None, 0,
                                               W.data_ptr(), 64, 64, None, ext.data_ptr(), aext.data_ptr(), None, 0,
                                               part.data_ptr(), _C.stream_ptr()))
W2 = torch.randn(64, 64, device=dev) * 0.1
Y2 = torch.empty(M, 64, device=dev); part2 = torch.empty(lib.s2c_rows_gemm_blocks(M, 64) * 2 * 64, device=dev)
report("plain (1M,64,64)", lambda: _C.call("s2c_rows_gemm", M, 64, 64, A.data_ptr(), 64, W2.data_ptr(), 64, None, None,
                                           Y2.data_ptr(), 64, part2.data_ptr(), _C.stream_ptr()))

# SA1's gather-fused first layer (random neighbour ids: the worst case for the gather)
_F = ctypes.c_float
_C.register("s2c_sa_gather_gemm", [_I, _I, _I, _I, _I, _L, _L, _F, _I, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P])
B, n, m, ns, C, Nout = 8, 40000, 2048, 64, 132, 64
pc = torch.randn(B, n, 3 + C, device=dev)
xyz = pc[..., :3].contiguous(); feats = pc[..., 3:]
inds = torch.stack([torch.randperm(n, device=dev)[:m] for _ in range(B)])
new_xyz = torch.gather(xyz, 1, inds.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
# neighbours of a centre: a run of nearby ids (ball-query rows are spatially coherent) / random ids
for label, idx in (("local", ((inds.view(B, m, 1) + torch.arange(ns, device=dev).view(1, 1, ns)) % n).to(torch.int32).contiguous()),
                   ("random", torch.randint(0, n, (B, m, ns), device=dev, dtype=torch.int32))):
    Wg = torch.randn(Nout, 3 + C, device=dev) * 0.1
    Mg = B * m * ns
    Yg = torch.empty(Mg, Nout, device=dev)
    pg = torch.empty(lib.s2c_rows_gemm_blocks(Mg, Nout) * 2 * Nout, device=dev)
    report("gather SA1 (%s ids)" % label,
           lambda: _C.call("s2c_sa_gather_gemm", B, n, m, ns, C, feats.stride(1), feats.stride(0), 0.2, 1,
                           xyz.data_ptr(), new_xyz.data_ptr(), feats.data_ptr(), idx.data_ptr(), Nout,
                           Wg.data_ptr(), 3 + C, Yg.data_ptr(), Nout, pg.data_ptr(), _C.stream_ptr()))
