"""Per-launch time of the decoder micro-kernels at the cfg3 shapes (R=8, E=300,
H=512, F=128, K=256): N back-to-back launches captured in a hipGraph."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd import _C
from scan2cap_amd.models import decoder_fused as df

dev = torch.device("cuda")
R, E, H, F, K = 8, 300, 512, 128, 256
g = torch.Generator(device="cuda").manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g) * 0.1
N = 200


def timed(name, fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s):
            for _ in range(N):
                fn()
        gr.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
    print("%-34s %7.2f us/launch" % (name, e0.elapsed_time(e1) * 1e3 / (5 * N)))


def lin(O, I, **kw):
    W, x, out = rn(O, I), rn(R, I), rn(R, O)
    b, a1 = rn(O), rn(R, O)
    return lambda: df._lin(R, O, I, W, I, x, I, out, O, bias=b, add1=a1, ld1=O, epi=1)


timed("lin O=300 I=512", lin(E, H))
timed("lin O=812 I=512", lin(H + E, H))
timed("lin O=300 I=128", lin(E, F))
timed("lin O=640 I=300", lin(F + H, E))
timed("lin O=512 I=512", lin(H, H))
timed("lin O=300 I=1536", lin(E, 3 * H))
timed("lin O=512 I=1536", lin(H, 3 * H))

W1, W2, x1, x2 = rn(E, 3 * H), rn(H, 3 * H), rn(R, 3 * H), rn(R, 3 * H)
o1, o2, gt, a1 = rn(R, E), rn(R, H), rn(R, E), rn(R, H)
timed("pair (300|512) I=1536", lambda: df._lin_pair(
    R, df._desc(E, 3 * H, W1, 3 * H, x1, 3 * H, o1, E, gate=gt, ldg=E, epi=2),
    df._desc(H, 3 * H, W2, 3 * H, x2, 3 * H, o2, H, add1=a1, ld1=H)))

Wih, Whh, bih, bhh = rn(3 * H, E), rn(3 * H, H), rn(3 * H), rn(3 * H)
x, h, hn = rn(R, E), rn(R, H), rn(R, H)
S = [rn(R, H) for _ in range(4)]
timed("gru_fwd", lambda: df._call("s2c_gru_fwd", R, H, E, df._p(Wih), df._p(Whh), df._p(bih),
                                  df._p(bhh), df._p(x), E, df._p(h), df._p(hn),
                                  df._p(S[0]), df._p(S[1]), df._p(S[2]), df._p(S[3])))
M, q, wa, mask, O = rn(R, K, H), rn(R, H + E), rn(H), torch.ones(R, K, device=dev), rn(R, K, F)
sc, al, att = rn(R, K), rn(R, K), rn(R, F)
timed("attn_fwd (2 kernels)", lambda: df._call("s2c_attn_fwd", R, K, H, F, df._p(M), df._p(q), H + E,
                                               df._p(wa), df._p(mask), df._p(O), df._p(sc),
                                               df._p(al), df._p(att), F))
dv, dM, dq, dwa = rn(R, F + H), rn(R, K, H), rn(R, H), rn(R, H)
timed("attn_bwd", lambda: df._call("s2c_attn_bwd", R, K, H, F, df._p(dv), F + H, df._p(att), F,
                                   df._p(al), df._p(O), df._p(M), df._p(q), H + E, df._p(wa),
                                   df._p(dM), df._p(dq), df._p(dwa)))
e = torch.empty(1, device=dev)
timed("torch add_ (1 elem)", lambda: e.add_(1.0))
