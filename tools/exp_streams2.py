import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd.pointnet2 import _ext
from scan2cap_amd.synthetic import scene_xyz
xyz = torch.from_numpy(scene_xyz(8, 40000)).cuda()
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
def main_work():
    c = a
    for _ in range(6): c = torch.mm(c, b) * 1e-3
    return c
def fps(): return _ext.furthest_point_sampling(xyz, 2048)
for _ in range(3): fps(); main_work()
torch.cuda.synchronize()
def t(fn, R=6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(R): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / R * 1e3
print("fps alone %.2f ms, main eager alone %.2f ms" % (t(fps), t(main_work)))
side = torch.cuda.Stream()
def both_eager():
    with torch.cuda.stream(side): fps()
    main_work()
print("eager main + side fps: %.2f ms" % t(both_eager))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    main_work()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g): out = main_work()
print("graph main alone %.2f ms" % t(g.replay))
def both_graph():
    with torch.cuda.stream(side): fps()
    g.replay()
print("graph main + side fps: %.2f ms" % t(both_graph))
hi = torch.cuda.Stream(priority=-1)
def both_prio():
    with torch.cuda.stream(hi): fps()
    g.replay()
print("graph main + HIGH-PRIORITY side fps: %.2f ms" % t(both_prio))
# CU-masked stream via hipExtStreamCreateWithCUMask
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(cu_lo, cu_hi, total=256):
    nwords = (total + 31) // 32
    mask = (ctypes.c_uint32 * nwords)()
    for cu in range(cu_lo, cu_hi):
        mask[cu // 32] |= (1 << (cu % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), nwords, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
def masked_stream_bits(bits, total=256):
    nwords = (total + 31) // 32
    mask = (ctypes.c_uint32 * nwords)()
    for cu in bits:
        mask[cu // 32] |= (1 << (cu % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), nwords, mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)
for per_xcd in (1, 2):
    side_bits = [32 * x + i for x in range(8) for i in range(per_xcd)]
    main_bits = [c for c in range(256) if c not in side_bits]
    side_m = masked_stream_bits(side_bits); main_m = masked_stream_bits(main_bits)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main_m):
        main_work(); torch.cuda.synchronize()
        with torch.cuda.graph(g2, stream=main_m): out2 = main_work()
    def main_masked_only():
        with torch.cuda.stream(main_m): g2.replay()
    def fps_masked_only():
        with torch.cuda.stream(side_m): fps()
    def both_masked():
        with torch.cuda.stream(side_m): fps()
        with torch.cuda.stream(main_m): g2.replay()
    print("per_xcd=%d: main masked alone %.2f ms | fps masked alone %.2f ms | both %.2f ms" % (
        per_xcd, t(main_masked_only), t(fps_masked_only), t(both_masked)))
