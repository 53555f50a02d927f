import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
hip = ctypes.CDLL("libamdhip64.so")
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
def main_work():
    c = a
    for _ in range(6): c = torch.mm(c, b) * 1e-3
    return c
def t(fn, R=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(R): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / R * 1e3
def mk(bits, nwords):
    mask = (ctypes.c_uint32 * nwords)()
    for cu in bits: mask[cu // 32] |= (1 << (cu % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), nwords, mask)
    assert rc == 0, rc
    got = (ctypes.c_uint32 * nwords)()
    hip.hipExtStreamGetCUMask(st, nwords, got)
    return torch.cuda.ExternalStream(st.value), [hex(x) for x in got]
print("default stream: %.2f ms" % t(main_work))
s0 = torch.cuda.Stream()
def on(s):
    def f():
        with torch.cuda.stream(s): main_work()
    return f
print("plain side stream: %.2f ms" % t(on(s0)))
allb = set(range(256))
cases = [
 ("remove 0..7", sorted(allb - set(range(8)))),
 ("remove 0..15", sorted(allb - set(range(16)))),
 ("remove 248..255", sorted(allb - set(range(248, 256)))),
 ("remove 240..255", sorted(allb - set(range(240, 256)))),
 ("only 0..7", list(range(8))),
 ("only 0..15", list(range(16))),
 ("only 248..255", list(range(248, 256))),
 ("remove bit 255", sorted(allb - {255})),
 ("remove odd>=240", sorted(allb - set(range(241, 256, 2)))),
]
for name, bits in cases:
    s_, got = mk(bits, 8)
    print("%-22s %.2f ms" % (name, t(on(s_))))
