"""Greedy decode of every proposal (models/caption_module.py:502-592) on the hand-written planes
GEMMs of csrc/s2c_planes.hip: per token SEVEN launches for all R = B*K rows --

    G1  x1 = relu(W_td[:, :E] emb[token] + t1)                              (caption_module.py:252-253)
    G2  h1 = GRUCell_1(x1, h1)         gates, biases and the cell in the GEMM's epilogue   (:254)
    G3  [q | l1] = [map_hidd ; W_lang[:, F:]] h1          one pass over h1 for both         (:257, :262)
    ATT alpha, att = local attention over the L gathered objects (csrc/s2c_decoder.hip)  (:257-261)
    G5  x2 = relu(W_lang[:, :F] att + l1 + b)                                                (:262)
    G6  h2 = GRUCell_2(x2, h2)                                                               (:263)
    G7  [logits | t1] = [classifier ; W_td[:, E:E+H]] h2 (+ P_tf): logits written in place, per-row
        arg-max keys, and the next token's map_topdown h2 block                (:553, :559-566, :252)

-- every product formed from bf16x3 planes (fp32-accurate, see the kernel), every activation
leaving its producer already split, the greedy feedback `embeddings[argmax]` resolved by G1's
prologue from G7's keys.  No library GEMM, no ATen kernel inside the token loop.
"""
import ctypes

import torch

from .. import _C

_I, _P, _LL = ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong


class _Seg(ctypes.Structure):
    """include/s2c_fused.h: s2c_planes_seg"""
    _fields_ = [("p", _P), ("pstride", _LL), ("ld", _I), ("kc", _I), ("rowmap", _P),
                ("rowdiv", _I), ("tiled", _I)]


class _GemmArgs(ctypes.Structure):
    """include/s2c_fused.h: s2c_planes_gemm_args"""
    _fields_ = [("M", _I), ("N", _I), ("gru", _I), ("relu", _I), ("nseg", _I), ("dbg", _I),
                ("seg", _Seg * 2), ("tokkeys", _P), ("ntokkeys", _I), ("ldw", _I), ("W", _P),
                ("wpstride", _LL), ("bias", _P), ("add", _P), ("C", _P), ("P", _P),
                ("ppstride", _LL), ("hprev", _P), ("amax", _P), ("ldadd", _I), ("ldc", _I),
                ("ldp", _I), ("ldh", _I), ("namax", _I), ("ptiled", _I), ("big_ok", _I),
                ("nsplit", _I), ("n1", _I), ("ldc2", _I), ("C2", _P)]


_C.register("s2c_planes_gemm", [_P, _P])
_C.register("s2c_planes_split", [_LL, _I, _P, _LL, _LL, _I, _P, _LL, _I, _P])
_C.register("s2c_attn_scene_fwd", [_I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P, ctypes.c_float, _P, _P, _I,
                                   _P, _LL, _I, _I, _P])
_C.register("s2c_attn_local_fwd_planes", [_I, _I, _I, _I, _P, _P, _I, _P, ctypes.c_float, _P, _P,
                                          _P, _P, _I, _P, _LL, _I, _I, _P])


DEBUG = 0      # tools/bench_planes.py: 1 = no products, 2 = no DMA (timing experiments only)


def set_big(mode):
    """256 x 256 tiles: -1 = where the grid still covers the chip (default), 0 never, 1 always."""
    lib = _C.load()
    lib.s2c_planes_set_big.argtypes = [_I]
    lib.s2c_planes_set_big.restype = None
    lib.s2c_planes_set_big(int(mode))


def _up(n, m):
    return (n + m - 1) // m * m


class Planes(object):
    """bf16x3 planes of a (rows x K) matrix: tensor t (3, rows_alloc, ld) bf16.  tiled (default):
    the 32-row x 16-column block layout the kernel streams (include/s2c_fused.h), rows allocated
    up to a multiple of 256 (the big tile reads whole 256-row groups); tiled=False: row-major, for operands gathered through a row map."""

    def __init__(self, rows, ld, device, zero=False, tiled=True):
        self.rows, self.ld, self.tiled = rows, ld, tiled
        self.rows_alloc = _up(rows, 256) if tiled else rows
        assert ld % (16 if tiled else 8) == 0
        self.t = (torch.zeros if zero else torch.empty)((3, self.rows_alloc, ld),
                                                        dtype=torch.bfloat16, device=device)
        self.pstride = self.rows_alloc * ld

    def ptr(self):
        return self.t.data_ptr()

    def dense(self):
        """(3, rows, ld) row-major view / copy of the planes (tests)."""
        if not self.tiled:
            return self.t[:, :self.rows]
        t = self.t.view(3, self.rows_alloc // 32, self.ld // 16, 32, 2, 8)
        r8 = (torch.arange(32, device=t.device) >> 3) & 1
        # slot s of row r32 holds half s ^ r8: undo the swap
        sw = torch.where(r8.view(32, 1, 1).bool(), t.flip(4), t)
        out = sw.permute(0, 1, 3, 2, 4, 5).reshape(3, self.rows_alloc, self.ld)
        return out[:, :self.rows]


def split(x, rows_out=None, ld=None, tiled=True):
    """fp32 (rows, K) [unit column stride] -> Planes, zero padded to (rows_out, ld)."""
    assert x.dim() == 2 and x.dtype == torch.float32 and x.is_cuda
    if x.stride(1) != 1:
        x = x.contiguous()
    rows, K = x.shape
    rows_out = rows if rows_out is None else rows_out
    ld = _up(K, 32) if ld is None else ld
    out = Planes(rows_out, ld, x.device, tiled=tiled)
    with torch.cuda.device(x.device):
        _C.call("s2c_planes_split", rows, K, x.data_ptr(), x.stride(0), out.rows_alloc, ld,
                out.ptr(), out.pstride, int(tiled), _C.stream_ptr())
    return out


def gemm(M, N, segs, W, bias=None, add=None, relu=False, C=None, P=None, gru=False, hprev=None,
         amax=None, tokkeys=None, split=None):
    """One s2c_planes_gemm launch.  segs: list of (Planes, kc[, rowmap tensor | rowdiv int]).
    split = (nsplit, n1, C2): two outputs side by side -- W rows [0, nsplit) -> C (n1 valid columns;
    bias, amax, P), W rows [nsplit, N) -> C2 (+ add)."""
    a = _GemmArgs()
    a.M, a.N, a.gru, a.relu, a.nseg, a.dbg = M, N, int(gru), int(relu), len(segs), DEBUG
    kct = 0
    big_ok = True
    for i, sg in enumerate(segs):
        pl, kc = sg[0], sg[1]
        a.seg[i].p, a.seg[i].pstride, a.seg[i].ld, a.seg[i].kc = pl.ptr(), pl.pstride, pl.ld, kc
        a.seg[i].tiled = int(pl.tiled)
        assert 32 * kc <= pl.ld
        mapped = (len(sg) > 2 and sg[2] is not None) or (i == 0 and tokkeys is not None)
        assert not (pl.tiled and mapped), "a gathered operand must be row-major planes"
        assert mapped or pl.rows >= M
        big_ok = big_ok and (mapped or pl.rows_alloc >= _up(M, 256))
        if len(sg) > 2 and sg[2] is not None:
            if torch.is_tensor(sg[2]):
                assert sg[2].dtype == torch.int32 and sg[2].is_contiguous()
                a.seg[i].rowmap = sg[2].data_ptr()
            else:
                a.seg[i].rowdiv = int(sg[2])
        kct += kc
    assert W.ld == 32 * kct, (W.ld, kct)
    assert W.tiled and W.rows_alloc >= (4 * N if gru else _up(N, 128))
    a.W, a.wpstride, a.ldw = W.ptr(), W.pstride, W.ld
    a.big_ok = int(big_ok and W.rows_alloc >= (_up(N, 64) * 4 if gru else _up(N, 256)))
    if tokkeys is not None:
        a.tokkeys, a.ntokkeys = tokkeys.data_ptr(), tokkeys.shape[1]
    if bias is not None:
        assert bias.is_contiguous() and bias.dtype == torch.float32
        a.bias = bias.data_ptr()
    if add is not None:
        assert add.stride(1) == 1
        a.add, a.ldadd = add.data_ptr(), add.stride(0)
    if C is not None:
        assert C.stride(1) == 1 and C.dtype == torch.float32
        a.C, a.ldc = C.data_ptr(), C.stride(0)
    if P is not None:
        a.P, a.ppstride, a.ldp, a.ptiled = P.ptr(), P.pstride, P.ld, int(P.tiled)
        assert P.rows >= M
    if hprev is not None:
        a.hprev, a.ldh = hprev.data_ptr(), hprev.stride(0)
    if amax is not None:
        a.amax, a.namax = amax.data_ptr(), amax.shape[1]
    if split is not None:
        a.nsplit, a.n1 = int(split[0]), int(split[1])
        assert split[2].stride(1) == 1 and split[2].dtype == torch.float32
        a.C2, a.ldc2 = split[2].data_ptr(), split[2].stride(0)
    if _C.TIMER.enabled:
        k = 32 * kct                                   # (K as padded to whole chunks)
        cols = 3 * N if gru else (split[1] + N - split[0] if split is not None else N)
        _C.TIMER.alg_flops = 2.0 * M * cols * k
        _C.TIMER.alg_bytes = 6 * (M * k + (4 * N if gru else N) * k) + 4 * M * N
    _C.call("s2c_planes_gemm", ctypes.byref(a), _C.stream_ptr())


def pack_gru(cell, Ep):
    """GRUCell weights -> (4H, Ep + H) rows grouped per 32 hidden units as [r | z | n_i | n_h]
    (the n_i rows multiply x only, the n_h rows h only), bias (4, H)."""
    W_ih, W_hh = cell.weight_ih.detach(), cell.weight_hh.detach()
    H, E = W_hh.shape[1], W_ih.shape[1]
    assert H % 32 == 0
    W = torch.zeros(4, H, Ep + H, device=W_ih.device)
    W[0, :, :E], W[0, :, Ep:] = W_ih[:H], W_hh[:H]
    W[1, :, :E], W[1, :, Ep:] = W_ih[H:2 * H], W_hh[H:2 * H]
    W[2, :, :E] = W_ih[2 * H:]
    W[3, :, Ep:] = W_hh[2 * H:]
    W = W.view(4, H // 32, 32, Ep + H).permute(1, 0, 2, 3).reshape(4 * H, Ep + H)
    b_ih, b_hh = cell.bias_ih.detach(), cell.bias_hh.detach()
    bias = torch.stack([b_ih[:H] + b_hh[:H], b_ih[H:2 * H] + b_hh[H:2 * H], b_ih[2 * H:],
                        b_hh[2 * H:]]).contiguous()
    return split(W.contiguous(), rows_out=4 * H), bias


def supported(mod, L, dense=False):
    """L: attended objects per row (num_locals), or -- dense -- the K objects of a scene."""
    return (mod.hidden_size % 32 == 0 and mod.feat_size % 4 == 0 and L <= (512 if dense else 32)
            and mod.attend.bias is None and mod.map_topdown[0].bias is not None)


def _weights(mod):
    """The planes of every weight of the step, split on EVERY decode() call: ~20 small launches
    against 29 x 7 per caption batch.  No cache -- a hipGraph replay of optimizer.step(), writes
    through `p.data` and EMA swaps change a weight without changing `data_ptr()` or `_version`,
    and stale planes would decode silently with the old weights
    (tests/test_planes_gpu.py::test_greedy_decode_sees_weights_changed_in_place).  Inside a
    captured evaluation step the split launches are part of the graph and re-run per replay."""
    ps = [mod.map_topdown[0].weight]
    E, H, F_ = mod.emb_size, mod.hidden_size, mod.feat_size
    Ep, Fp = _up(E, 32), _up(F_, 32)
    dev = ps[0].device
    with torch.no_grad():
        W_td = mod.map_topdown[0].weight.detach()
        W_lang = mod.map_lang[0].weight.detach()
        V = mod.classifier.weight.shape[0]
        VS, HS = _up(V, 128), _up(H, 128)
        w = {"Ep": Ep, "Fp": Fp, "VS": VS, "HS": HS}
        # every product that reads h2 in ONE pass over its planes, likewise h1 (two outputs side by
        # side, s2c_planes_gemm `nsplit`): [classifier ; map_topdown's h2 block], [map_hidd ; map_lang's
        # h1 block] -- the word / attention products keep only their own short K
        w["W1x"] = split(W_td[:, :E], rows_out=_up(E, 128), ld=Ep)
        Wc2 = torch.zeros(VS + E, H, device=dev)
        Wc2[:V], Wc2[VS:] = mod.classifier.weight.detach(), W_td[:, E:E + H]
        w["Wc2"] = split(Wc2, rows_out=_up(VS + E, 128))
        Wq2 = torch.zeros(HS + E, H, device=dev)
        Wq2[:H], Wq2[HS:] = mod.map_hidd.weight.detach(), W_lang[:, F_:]
        w["Wq2"] = split(Wq2, rows_out=_up(HS + E, 128))
        w["W5a"] = split(W_lang[:, :F_], rows_out=_up(E, 128), ld=Fp)
        w["Wtf"] = split(W_td[:, E + H:], rows_out=_up(E, 128), ld=Fp)
        w["b_td"] = mod.map_topdown[0].bias.detach().contiguous()
        w["Wg1"], w["bg1"] = pack_gru(mod.recurrent_cell_1, Ep)
        w["Wm"] = split(mod.map_feat.weight.detach(), rows_out=_up(H, 128), ld=Fp)
        w["wa"] = mod.attend.weight.detach().reshape(-1).contiguous()
        w["b_lang"] = mod.map_lang[0].bias.detach().contiguous()
        w["Wg2"], w["bg2"] = pack_gru(mod.recurrent_cell_2, Ep)
        w["b_cls"] = mod.classifier.bias.detach().contiguous()
        w["emb"] = split(mod._emb_table.detach(), ld=Ep, tiled=False)     # gathered by token
    return w


def decode(mod, sos, rows_per_scene, target_feats, local, T, scene_valid=None):
    """sos (B,E): the first input word of every scene (lang_feat[:, 0]); target_feats (R,F);
    local (R,L,F): the attended objects of every row (relation features added).
    Returns logits (T,R,V) step-major and alpha (T,R,L).
    scene_valid (B,K) float 0/1: num_locals = -1 -- every row attends over ALL K objects of its
    scene; `local` is then the scene's objects (B,K,F) themselves, shared by the scene's rows
    (s2c_attn_scene_fwd), and alpha is (T,R,K)."""
    dev = local.device
    dense = scene_valid is not None
    if dense:
        Bs, L, F_ = local.shape
        R = target_feats.shape[0]
        assert R == Bs * rows_per_scene
    else:
        R, L, F_ = local.shape
    E, H, V = mod.emb_size, mod.hidden_size, mod.num_vocabs
    with torch.cuda.device(dev):
        w = _weights(mod)
        Ep, Fp = w["Ep"], w["Fp"]
        VS, HS = w["VS"], w["HS"]
        nct = VS // 128
        sos_p = split(sos.contiguous(), ld=Ep, tiled=False)             # gathered: row r / K
        tf_p = split(target_feats.contiguous(), ld=Fp)
        P_tf = torch.empty(R, E, device=dev)
        gemm(R, E, [(tf_p, Fp // 32)], w["Wtf"], bias=w["b_td"], C=P_tf)
        local_c = local.contiguous()
        NL = local_c.shape[0] * L                       # mapped rows: R L, or B K (scene-shared keys)
        loc_p = split(local_c.view(NL, F_), ld=Fp)
        mapped = torch.empty(NL, H, device=dev)
        gemm(NL, H, [(loc_p, Fp // 32)], w["Wm"], C=mapped)
        if dense:
            valid_c = scene_valid.to(torch.float32).contiguous()
        h1 = torch.zeros(2, R, H, device=dev)
        h2 = torch.zeros(2, R, H, device=dev)
        h1p = [Planes(R, H, dev, zero=True), Planes(R, H, dev)]
        h2p = [Planes(R, H, dev, zero=True), Planes(R, H, dev)]
        x1p, x2p = Planes(R, Ep, dev), Planes(R, Ep, dev)
        attp = Planes(R, Fp, dev, zero=Fp != F_)
        qh = torch.empty(R, H, device=dev)
        t1 = P_tf.clone()       # map_topdown's h2 block + target block + bias: h2 = 0 before the first word
        l1 = torch.empty(R, E, device=dev)                              # map_lang's h1 block
        keys = torch.empty(R, nct, dtype=torch.int64, device=dev)
        cap = torch.empty(T, R, V, device=dev)
        alpha = torch.empty(T, R, L, device=dev)
        st = _C.stream_ptr()
        cur = 0
        for t in range(T):
            nxt = cur ^ 1
            word = (sos_p, Ep // 32, rows_per_scene) if t == 0 else (w["emb"], Ep // 32)
            gemm(R, E, [word], w["W1x"], add=t1, relu=True, P=x1p,
                 tokkeys=None if t == 0 else keys)
            gemm(R, H, [(x1p, Ep // 32), (h1p[cur], H // 32)], w["Wg1"], bias=w["bg1"], gru=True,
                 hprev=h1[cur], C=h1[nxt], P=h1p[nxt])
            gemm(R, HS + E, [(h1p[nxt], H // 32)], w["Wq2"], C=qh, split=(HS, H, l1))
            if dense:
                if _C.TIMER.enabled:  # the scene's keys once per row block + the rows' q, alpha, att
                    _C.TIMER.alg_bytes = 4 * (NL * (H + F_) + R * (H + L + F_))
                _C.call("s2c_attn_scene_fwd", R, rows_per_scene, L, H, F_, mapped.data_ptr(),
                        valid_c.data_ptr(), local_c.data_ptr(), qh.data_ptr(), H, w["wa"].data_ptr(),
                        0.0, alpha[t].data_ptr(), None, F_, attp.ptr(), attp.pstride, Fp,
                        int(attp.tiled), st)
            else:
                if _C.TIMER.enabled:      # one pass over mapped + local features
                    _C.TIMER.alg_bytes = 4 * (R * L * (H + F_ + 1) + R * (H + F_))
                _C.call("s2c_attn_local_fwd_planes", R, L, H, F_, mapped.data_ptr(), qh.data_ptr(), H,
                        w["wa"].data_ptr(), 0.0, None, local_c.data_ptr(), alpha[t].data_ptr(), None,
                        F_, attp.ptr(), attp.pstride, Fp, int(attp.tiled), st)
            gemm(R, E, [(attp, Fp // 32)], w["W5a"], bias=w["b_lang"], add=l1, relu=True, P=x2p)
            gemm(R, H, [(x2p, Ep // 32), (h2p[cur], H // 32)], w["Wg2"], bias=w["bg2"], gru=True,
                 hprev=h2[cur], C=h2[nxt], P=h2p[nxt])
            gemm(R, VS + E, [(h2p[nxt], H // 32)], w["Wc2"], bias=w["b_cls"], C=cap[t], amax=keys,
                 split=(VS, V, t1), add=P_tf)
            cur = nxt
    return cap, alpha
