"""Streaming-copy probes (tools/probes/s2c_probe.hip): the bandwidth ceiling of a (M x 64) fp32
row pass with plain loads vs a wave-private LDS-DMA ring."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd import _C
from tools.bench_ops import timeit
from scan2cap_amd import build as _b
lib = ctypes.CDLL(_b.build_probes())
_I, _L, _P = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p
lib.s2c_probe_copy.argtypes = [_I, _L, _P, _P, _I, _I, _P]
M = 1048576
A = torch.randn(M * 64 + 64, device="cuda"); Y = torch.empty(M * 64, device="cuda")
for mode, blocks in [(0, 2048), (0, 8192), (0, 32768), (1, 256), (1, 240), (1, 512)]:
    for mis in ((0,) if mode == 0 else (0, 1)):
        f = lambda: lib.s2c_probe_copy(mode, M, A.data_ptr(), Y.data_ptr(), blocks, mis, torch.cuda.current_stream().cuda_stream)
        Y.zero_(); rc = f(); torch.cuda.synchronize()
        ok = torch.equal(Y.view(M, 64)[:, 1:], A[mis:mis + M * 64].view(M, 64)[:, 1:]) and \
            torch.allclose(Y.view(M, 64)[:, ::4], A[mis:mis + M * 64].view(M, 64)[:, ::4] + 1)
        t = timeit(f, iters=20)
        print("mode %d blocks %5d misalign %d: rc %d ok %s  %.1f us  %.2f TB/s" % (mode, blocks, mis, rc, ok, t, 2 * M * 256 / t / 1e6))
