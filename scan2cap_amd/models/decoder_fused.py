"""Teacher-forced top-down decoder as ONE autograd Function over the hand-written
small-batch kernels of csrc/s2c_decoder.hip (C ABI: include/s2c_fused.h).

Computes exactly TopDownSceneCaptionModule._step (models/caption_module.py:250-292)
for `steps` sequential steps and its back-propagation through time, but:
  * per step: 7 forward / 6 backward kernel launches instead of ~25 / ~50;
  * everything that does not depend on the recurrence is hoisted into a handful of
    large GEMMs outside the loop: the word and target-feature columns of
    `map_topdown`, `map_feat(obj_feats)`, the classifier over all steps, and EVERY
    weight gradient (sum_t delta_t^T a_t == one stacked GEMM);
  * no host synchronisation, fixed shapes => hipGraph-capturable.
Splitting `map_topdown` / `map_lang` by column blocks changes the fp32 summation
order only (|diff| ~1e-6, tolerance 1e-4).
"""
import ctypes

import torch
from torch.autograd import Function

from .. import _C
from ..pointnet2 import fused

_I, _P = ctypes.c_int, ctypes.c_void_p
_C.register("s2c_small_linear", [_I, _I, _I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _P, _I, _P])
_C.register("s2c_gru_fwd", [_I, _I, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P])
_C.register("s2c_gru_gates_bwd", [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P])
_C.register("s2c_attn_fwd", [_I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P])
_C.register("s2c_attn_bwd", [_I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P])
_C.register("s2c_small_linear_pair", [_I, _P, _P, _P, _P])


class _LinDesc(ctypes.Structure):
    """include/s2c_fused.h: s2c_lin_desc"""
    _fields_ = ([(n, _P) for n in ("W", "x", "bias", "add1", "add2", "gate", "out")] +
                [(n, _I) for n in ("O", "I", "ldw", "ldx", "ld1", "ld2", "ldg", "ldo", "epi")])


class _GruBwdDesc(ctypes.Structure):
    """include/s2c_fused.h: s2c_gru_bwd_desc"""
    _fields_ = [(n, _P) for n in ("sr", "sz", "sn", "sghn", "hprev", "dgi", "dgh",
                                  "dh_direct")]


def _p(t):
    return t.data_ptr() if t is not None else None


def _call(name, *args, alg_bytes=0):
    if _C.TIMER.enabled:
        _C.TIMER.alg_bytes = int(alg_bytes)
    _C.call(name, *args, _C.stream_ptr())


def _lin(R, O, I, W, ldw, x, ldx, out, ldo, bias=None, add1=None, ld1=0, add2=None,
         ld2=0, gate=None, ldg=0, epi=0):
    # algorithmic bytes: the weight block is streamed once, plus the small operands
    _call("s2c_small_linear", R, O, I, _p(W), ldw, _p(x), ldx, _p(bias), _p(add1), ld1,
          _p(add2), ld2, _p(gate), ldg, epi, _p(out), ldo,
          alg_bytes=4 * (O * I + R * I + 2 * R * O))


def _desc(O, I, W, ldw, x, ldx, out, ldo, bias=None, add1=None, ld1=0, add2=None, ld2=0,
          gate=None, ldg=0, epi=0):
    return _LinDesc(_p(W), _p(x), _p(bias), _p(add1), _p(add2), _p(gate), _p(out),
                    O, I, ldw, ldx, ld1, ld2, ldg, ldo, epi)


def _gates(S, t, hprev, dgi, dgh, dh_direct):
    return _GruBwdDesc(_p(S[0][t]), _p(S[1][t]), _p(S[2][t]), _p(S[3][t]), _p(hprev),
                       _p(dgi), _p(dgh), _p(dh_direct))


def _lin_pair(R, d1, d2=None, gates=None):
    ab = 4 * (d1.O * d1.I + R * d1.I + 2 * R * d1.O)
    if d2 is not None:
        ab += 4 * (d2.O * d2.I + R * d2.I + 2 * R * d2.O)
    if gates is not None:
        ab += 4 * 12 * R * d1.O
    _call("s2c_small_linear_pair", R, ctypes.addressof(d1),
          ctypes.addressof(d2) if d2 is not None else None,
          ctypes.addressof(gates) if gates is not None else None, alg_bytes=ab)


ENABLED = True      # scan2cap_amd/opbyop.py: False = the plain `_step` loop
# One-pass attention (attn_local_kernel, one wave per row) for K <= this many keys.  0 = off:
# measured at cfg3 (R = 8 rows, K = 10 gathered objects) the single launch is SLOWER than the
# scores + softmax pair (10.49 vs 10.40 ms per step): one wave per row leaves 8 waves on the
# chip for 10 x 512 tanh each, the pair spreads them over 80 + 32 workgroups.  (The greedy
# decoder's thousands of rows are where the one-pass kernel pays.)  Limit of the kernel: 32.
LOCAL_ATTN_MAX_K = 0
_C.register("s2c_attn_local_fwd", [_I, _I, _I, _I, _P, _P, _I, _P, ctypes.c_float, _P, _P, _P,
                                   _P, _I, _P])
# few keys: scores + softmax + weighted sum + the map_lang layer in ONE launch (attn_x2_kernel:
# 7 -> 5 dependent launches per forward step); = False: the three launches
import os as _os
FUSE_ATTN_X2 = True
ATTN_X2_MAX_K = 32
_C.register("s2c_attn_x2_fwd", [_I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P,
                                _P, _I, _P, _I, _P])
# ... and its backward mirror: the transposed map_lang product inside the attention backward
# (6 -> 5 dependent launches per backward step); = False: the two launches
FUSE_ATTN_X2_BWD = True
_C.register("s2c_attn_bwd_x2", [_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _I, _P, _P,
                                _P, _I, _P, _P])


# The hoisted products (in front of / behind the step loop, and every weight gradient) as a few
# multi-job launches of csrc/s2c_mgemm.hip instead of ~26 library GEMM calls of 6-48 us each;
# = False: torch.mm / matmul / bmm (hipBLASLt), the reference formulation the tests compare with
from .. import mgemm as _mg
USE_MGEMM = True
# the three classifier products of the teacher-forced decoder on s2c_small_gemm_ex (bf16x3 MFMA)
MFMA_CLASSIFIER = True


# The whole forward recurrence as ONE persistent kernel (csrc/s2c_decoder_persist.hip): 128
# co-resident workgroups exchange the per-step vectors as tagged values instead of meeting at
# 5 T launch boundaries.  S2C_DECODER_PERSIST=0 (or set_persist(False)): the launch chain.
class _DecFwdArgs(ctypes.Structure):
    """include/s2c_fused.h: s2c_dec_fwd_args"""
    _fields_ = ([(n, _I) for n in ("R", "K", "H", "E", "F", "T", "ldtd", "ldlang", "backoff",
                                   "pad_")] +
                [(n, _P) for n in ("W_td_h2", "Pw", "Ptf", "W_ih1", "W_hh1", "b_ih1", "b_hh1",
                                   "Wqh", "M", "wa", "mask", "O", "W_lang", "b_lang", "W_ih2",
                                   "W_hh2", "b_ih2", "b_hh2", "H1", "H2", "X1", "X2", "S", "C",
                                   "QL", "ALPHA", "ATT", "xbuf", "prof", "nonce", "started",
                                   "fail")])


class _DecBwdArgs(ctypes.Structure):
    """include/s2c_fused.h: s2c_dec_bwd_args"""
    _fields_ = ([(n, _I) for n in ("R", "K", "H", "E", "T", "pad_")] +
                [(n, _P) for n in ("dH2", "C", "S", "X1", "X2", "QL", "ALPHA", "M", "wa", "P",
                                   "Latt", "WT_ih2", "WT_hh2", "WT_hl", "WT_ih1", "WT_hh1",
                                   "WT_td", "DA1", "DQA", "DG", "dM", "dwa_rows", "xbuf", "prof",
                                   "nonce", "started", "fail")])


_C.register("s2c_decoder_bwd_persist", [_P, _P])


_C.register("s2c_decoder_fwd_persist", [_P, _P])
PERSIST_BACKOFF = 0
PROF_BWD = None
PROF = None     # tools/bench_decoder_persist.py: an int64 tensor (8 * T * 16) of phase stamps
_XBUF = {}      # (device index, H, E) -> (exchange buffer, [nonce, started]); zeroed once


def _plib():
    lib = _C.load()
    if not getattr(lib, "_s2c_persist_typed", False):
        lib.s2c_decoder_fwd_persist_supported.argtypes = [_I] * 6
        lib.s2c_decoder_fwd_persist_supported.restype = _I
        lib.s2c_decoder_fwd_persist_xbuf_pairs.argtypes = [_I, _I]
        lib.s2c_decoder_fwd_persist_xbuf_pairs.restype = ctypes.c_longlong
        lib.s2c_decoder_bwd_persist_supported.argtypes = [_I] * 5
        lib.s2c_decoder_bwd_persist_supported.restype = _I
        lib.s2c_decoder_bwd_persist_xbuf_pairs.argtypes = [_I, _I]
        lib.s2c_decoder_bwd_persist_xbuf_pairs.restype = ctypes.c_longlong
        lib.s2c_decoder_persist_set.argtypes = [_I]
        lib.s2c_decoder_persist_set.restype = None
        lib._s2c_persist_typed = True
    return lib


def persist_failed(dev=None):
    """True if a launch of the persistent kernel gave up on a poll (its results are invalid).
    Synchronises; for tests, smoke() and training loops (check it once per step or per epoch,
    outside any capture: the NaN poison of the kernels is the loud-by-default signal, this flag is
    the reliable one)."""
    return any(int(ctl[2].item()) != 0 for k, (_, ctl) in _XBUF.items()
               if dev is None or k[0] == torch.device(dev).index)


def set_persist(on):
    _plib().s2c_decoder_persist_set(int(bool(on)))


# The persistent kernels spin until all of their workgroups are co-resident.  They are ordinary
# launches (an occupancy query sizes the grid), so NOTHING guarantees forward progress when a
# second process holds CUs of the same device: each process takes an exclusive advisory lock per
# device before its first persistent launch and silently uses the launch chain when another
# process already holds it (two trainers, or an evaluation job beside a trainer, on one GPU).
_DEVICE_LOCKS = {}      # device index -> open file object (lock held) or False (use the chain)


def _device_lock_path(index):
    """Per-user directory (mode 0700) under the temp dir: another user's lock file of the same
    name can neither block this process nor be followed through a symlink."""
    try:
        ident = str(torch.cuda.get_device_properties(index).uuid)
    except Exception:
        ident = "%s-%d" % (_os.environ.get("HIP_VISIBLE_DEVICES",
                                           _os.environ.get("ROCR_VISIBLE_DEVICES", "all")), index)
    ident = "".join(c if c.isalnum() else "_" for c in ident)
    import tempfile
    d = _os.path.join(tempfile.gettempdir(), "s2c-%d" % _os.getuid())
    _os.makedirs(d, mode=0o700, exist_ok=True)
    return _os.path.join(d, "persist_%s.lock" % ident)


def _open_lock(path):
    """O_NOFOLLOW (no symlink games in a shared temp dir), O_CLOEXEC (an exec'd child does not
    inherit the descriptor; forked workers are handled by the at-fork hook below)."""
    fd = _os.open(path, _os.O_RDWR | _os.O_CREAT | _os.O_NOFOLLOW | _os.O_CLOEXEC, 0o600)
    return _os.fdopen(fd, "r+")


_WARNED_LOCK = []


def persist_allowed(dev):
    """This process may run the persistent decoder kernels on `dev` (it owns the device's lock).
    `parallel.init_from_env` additionally switches them off when ranks share a device."""
    index = torch.device(dev).index
    if index is None:
        index = torch.cuda.current_device()
    got = _DEVICE_LOCKS.get(index)
    if got is None:
        got = False
        if _os.environ.get("S2C_PERSIST_LOCK", "1") == "0":
            got = True
        else:
            try:
                import fcntl
                f = _open_lock(_device_lock_path(index))
                try:
                    fcntl.flock(f, fcntl.LOCK_EX | fcntl.LOCK_NB)
                    got = f                     # held until the process exits
                except OSError:
                    f.close()                   # another process owns the device: launch chain
            except Exception as e:              # no lock file possible: be conservative, say so once
                got = False
                if not _WARNED_LOCK:
                    _WARNED_LOCK.append(1)
                    import warnings
                    warnings.warn("scan2cap_amd: the per-device lock of the persistent decoder "
                                  "kernels cannot be taken (%r); the decoder runs as a launch "
                                  "chain in this process" % (e,))
        _DEVICE_LOCKS[index] = got
    return got is not False


def _drop_locks_in_child():
    # a forked child (DataLoader worker) shares the open file description, hence the flock: it must
    # not keep the device reserved after the parent exits, and must not launch persistently itself
    for index, f in list(_DEVICE_LOCKS.items()):
        if f not in (True, False):
            try:
                # close THROUGH the file object (os.close(f.fileno()) would let its finalizer close
                # the same descriptor number a second time: EBADF noise, or an unrelated descriptor
                # if the number was reused).  The flock belongs to the open file description the
                # parent still holds: closing the child's duplicate does not release it.
                f.close()
            except Exception:
                pass
        _DEVICE_LOCKS[index] = False


if hasattr(_os, "register_at_fork"):
    _os.register_at_fork(after_in_child=_drop_locks_in_child)


def release_device_locks():
    """Give the per-device locks back (a process that is done with its persistent launches and
    hands the device to a child: bench.py's builder-fed run).  The next persistent launch of this
    process takes them again."""
    for index, f in list(_DEVICE_LOCKS.items()):
        if f not in (True, False):
            try:
                f.close()
            except Exception:
                pass
        del _DEVICE_LOCKS[index]


_LAST_STREAM = {}       # scratch key -> the stream of its previous launch


def _persist_scratch(dev, H, E, bwd=False):
    # one exchange buffer + nonce per (device, kernel), allocated ONCE and never inside a capture
    # (the zero-fill would be replayed with every step and land in the graph's private pool).
    # Two launches that share the scratch must not overlap (same nonce -> same tags): a launch on
    # another stream than the previous one first waits for that stream.
    key = (dev.index, H, E, bwd)
    if key not in _XBUF:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("persistent decoder scratch must exist before a graph capture: run "
                               "the step once eagerly first (graphs.GraphedCallable does)")
        pairs = int(_plib().s2c_decoder_bwd_persist_xbuf_pairs(H, E) if bwd else
                    _plib().s2c_decoder_fwd_persist_xbuf_pairs(H, E))
        _XBUF[key] = (torch.zeros(pairs, dtype=torch.int64, device=dev),
                      torch.zeros(4, dtype=torch.int32, device=dev))
    cur = torch.cuda.current_stream(dev)
    last = _LAST_STREAM.get(key)
    if last is not None and last != cur and not torch.cuda.is_current_stream_capturing():
        cur.wait_stream(last)
    _LAST_STREAM[key] = cur
    return _XBUF[key]


def supported(emb, hid, feat, K):
    return (ENABLED and emb % 4 == 0 and hid % 4 == 0 and feat in (32, 64, 128, 256) and K <= 1024
            and hid <= 512 and emb <= 512)     # register-resident input slices


class _PrepArgs(ctypes.Structure):
    """s2c_prep_args (include/s2c_fused.h)."""
    _fields_ = [("n_transpose", ctypes.c_int), ("n_zero", ctypes.c_int),
                ("src", ctypes.c_void_p * 8), ("dst", ctypes.c_void_p * 8),
                ("rows", ctypes.c_int * 8), ("cols", ctypes.c_int * 8),
                ("lds", ctypes.c_longlong * 8), ("zero", ctypes.c_void_p * 8),
                ("zero_count", ctypes.c_longlong * 8)]


_C.register("s2c_batch_prep", [ctypes.c_void_p, ctypes.c_void_p])


def _batch_prep(srcs, dsts, zeros):
    a = _PrepArgs()
    a.n_transpose, a.n_zero = len(srcs), len(zeros)
    for j, (w, d) in enumerate(zip(srcs, dsts)):
        assert w.dim() == 2 and w.stride(1) == 1 and w.dtype == torch.float32
        a.src[j], a.dst[j] = w.data_ptr(), d.data_ptr()
        a.rows[j], a.cols[j], a.lds[j] = w.shape[0], w.shape[1], w.stride(0)
    for j, zt in enumerate(zeros):
        a.zero[j], a.zero_count[j] = zt.data_ptr(), zt.numel()
    _C.call("s2c_batch_prep", ctypes.byref(a), _C.stream_ptr())


class TopDownDecode(Function):
    """forward(word_embs (R,Tw,E), target_feats (R,F), obj_feats (R,K,F),
    masks (R,K) float, steps, *params) -> logits (R,steps,V), attn (R,K,steps).

    params (order): W_td, b_td, W_ih1, W_hh1, b_ih1, b_hh1, W_f, W_h, w_a, W_lang,
    b_lang, W_ih2, W_hh2, b_ih2, b_hh2, W_cls, b_cls."""

    @staticmethod
    def forward(ctx, word_embs, target_feats, obj_feats, masks, steps, *params):
        (W_td, b_td, W_ih1, W_hh1, b_ih1, b_hh1, W_f, W_h, w_a, W_lang, b_lang,
         W_ih2, W_hh2, b_ih2, b_hh2, W_cls, b_cls) = params
        dev = obj_feats.device
        R, K, F = obj_feats.shape
        E = W_td.shape[0]
        H = W_hh1.shape[1]
        T = int(steps)
        words = word_embs[:, :T].contiguous()                       # (R,T,E)
        O = obj_feats.contiguous()
        tf = target_feats.contiguous()
        mask = masks.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            # ---- hoisted, recurrence-free GEMMs -----------------------------
            if USE_MGEMM:
                Pw = torch.empty(R, T, E, device=dev)
                Ptf = torch.empty(R, E, device=dev)
                M = torch.empty(R, K, H, device=dev)
                _mg.launch([_mg.mm(words.view(R * T, E), W_td[:, :E].t(), Pw.view(R * T, E)),
                            _mg.mm(tf, W_td[:, E + H:].t(), Ptf, bias=b_td),
                            _mg.mm(O.view(R * K, F), W_f.t(), M.view(R * K, H))])
            else:
                Pw = torch.matmul(words, W_td[:, :E].t())               # (R,T,E)
                Ptf = torch.addmm(b_td, tf, W_td[:, E + H:].t())        # (R,E)
                M = torch.matmul(O, W_f.t())                            # (R,K,H)
            Wqh = torch.cat([W_h, W_lang[:, F:]], 0).contiguous()   # (H+E, H)
            wa = w_a.reshape(-1).contiguous()
            z = lambda *s: torch.zeros(*s, device=dev)
            e = lambda *s: torch.empty(*s, device=dev)
            H1, H2 = z(T + 1, R, H), z(T + 1, R, H)
            X1, X2 = e(T, R, E), e(T, R, E)
            S = e(2, 4, T, R, H)                  # r, z, n, gh_n of GRU 1, then of GRU 2
            S1, S2 = [S[0, j] for j in range(4)], [S[1, j] for j in range(4)]
            COEF = None
            QL = e(T, R, H + E)
            ALPHA, SC = e(T, R, K), e(R, K)
            ATT = e(T, R, F)
            ldtd, ldlang = W_td.shape[1], W_lang.shape[1]
            td_h2 = W_td[:, E:E + H]          # column block, row stride ldtd
            persist = (ldtd % 4 == 0 and ldlang % 4 == 0 and
                       _plib().s2c_decoder_fwd_persist_supported(R, K, H, E, F, T) == 1 and
                       persist_allowed(dev))
            if persist:
                xbuf, ctl = _persist_scratch(dev, H, E)
                if any(ctx.needs_input_grad) and \
                        _plib().s2c_decoder_bwd_persist_supported(R, K, H, E, T) == 1:
                    COEF = e(2, 4, T, R, H)       # gate-gradient coefficients for the backward kernel
                a = _DecFwdArgs()
                a.R, a.K, a.H, a.E, a.F, a.T, a.ldtd, a.ldlang = R, K, H, E, F, T, ldtd, ldlang
                for n, v in (("W_td_h2", td_h2), ("Pw", Pw), ("Ptf", Ptf), ("W_ih1", W_ih1),
                             ("W_hh1", W_hh1), ("b_ih1", b_ih1), ("b_hh1", b_hh1), ("Wqh", Wqh),
                             ("M", M), ("wa", wa), ("mask", mask), ("O", O), ("W_lang", W_lang),
                             ("b_lang", b_lang), ("W_ih2", W_ih2), ("W_hh2", W_hh2),
                             ("b_ih2", b_ih2), ("b_hh2", b_hh2), ("H1", H1), ("H2", H2),
                             ("X1", X1), ("X2", X2), ("QL", QL), ("ALPHA", ALPHA), ("ATT", ATT),
                             ("S", S), ("xbuf", xbuf)):
                    assert v.is_contiguous() or n == "W_td_h2", n
                    setattr(a, n, v.data_ptr())
                a.backoff = PERSIST_BACKOFF
                a.prof = PROF.data_ptr() if PROF is not None else None
                a.nonce, a.started, a.fail = ctl.data_ptr(), ctl.data_ptr() + 4, ctl.data_ptr() + 8
                a.C = COEF.data_ptr() if COEF is not None else None
                if _C.TIMER.enabled:     # weights once + everything saved for the backward pass
                    _C.TIMER.alg_bytes = 4 * (E * H + 6 * H * (E + H) + (H + E) * H + E * F +
                                              T * R * (11 * H + 3 * E + K + F))
                _C.call("s2c_decoder_fwd_persist", ctypes.byref(a), _C.stream_ptr())
            for t in range(0 if not persist else T, T):
                _lin(R, E, H, td_h2, ldtd, H2[t], H, X1[t], E, add1=Pw[:, t],
                     ld1=T * E, add2=Ptf, ld2=E, epi=1)
                _call("s2c_gru_fwd", R, H, E, _p(W_ih1), _p(W_hh1), _p(b_ih1),
                      _p(b_hh1), _p(X1[t]), E, _p(H1[t]), _p(H1[t + 1]),
                      _p(S1[0][t]), _p(S1[1][t]), _p(S1[2][t]), _p(S1[3][t]),
                      alg_bytes=4 * (3 * H * (E + H) + R * (E + 6 * H)))
                _lin(R, H + E, H, Wqh, H, H1[t + 1], H, QL[t], H + E)
                if FUSE_ATTN_X2 and K <= ATTN_X2_MAX_K and F % 4 == 0 and F <= 256 \
                        and ldlang % 4 == 0:
                    _call("s2c_attn_x2_fwd", R, K, H, F, E, _p(M), _p(QL[t]), H + E, _p(wa),
                          _p(mask), _p(O), _p(W_lang), ldlang, _p(b_lang), _p(QL[t][:, H:]),
                          H + E, _p(ALPHA[t]), _p(ATT[t]), F, _p(X2[t]), E,
                          alg_bytes=4 * (R * K * (H + F) + E * F + R * (H + 2 * K + F + 2 * E)))
                    _call("s2c_gru_fwd", R, H, E, _p(W_ih2), _p(W_hh2), _p(b_ih2),
                          _p(b_hh2), _p(X2[t]), E, _p(H2[t]), _p(H2[t + 1]),
                          _p(S2[0][t]), _p(S2[1][t]), _p(S2[2][t]), _p(S2[3][t]),
                          alg_bytes=4 * (3 * H * (E + H) + R * (E + 6 * H)))
                    continue
                if K <= LOCAL_ATTN_MAX_K:
                    # few keys (the num_locals gather): scores, mask, softmax and the weighted
                    # sum in ONE pass, one wave per row (attn_local_kernel)
                    _call("s2c_attn_local_fwd", R, K, H, F, _p(M), _p(QL[t]), H + E, _p(wa),
                          0.0, _p(mask), _p(O), _p(ALPHA[t]), _p(ATT[t]), F,
                          alg_bytes=4 * (R * K * (H + F) + R * (H + 2 * K + F)))
                else:
                    _call("s2c_attn_fwd", R, K, H, F, _p(M), _p(QL[t]), H + E, _p(wa),
                          _p(mask), _p(O), _p(SC), _p(ALPHA[t]), _p(ATT[t]), F,
                          alg_bytes=4 * (R * K * (H + F) + R * (H + 2 * K + F)))
                _lin(R, E, F, W_lang, ldlang, ATT[t], F, X2[t], E, bias=b_lang,
                     add1=QL[t][:, H:], ld1=H + E, epi=1)
                _call("s2c_gru_fwd", R, H, E, _p(W_ih2), _p(W_hh2), _p(b_ih2),
                      _p(b_hh2), _p(X2[t]), E, _p(H2[t]), _p(H2[t + 1]),
                      _p(S2[0][t]), _p(S2[1][t]), _p(S2[2][t]), _p(S2[3][t]),
                      alg_bytes=4 * (3 * H * (E + H) + R * (E + 6 * H)))
            RH = R * H
            if USE_MGEMM:
                # rows (r, t) of the logits read H2[t + 1, r] in place: no (R,T,H) copy
                H2n = None
                V = W_cls.shape[0]
                logits = torch.empty(R, T, V, device=dev)
                # the classifier (R T x V x H) on the matrix cores (csrc/s2c_sgemm.hip); layouts it does
                # not take stay on the VALU multi-GEMM
                if not (MFMA_CLASSIFIER and _mg.mfma(_mg.NT, R * T, V, H, H2[1:],
                                                     _mg.ax(RH, div=T, hi=H), W_cls, _mg.ax(H),
                                                     logits, _mg.ax(V), bias=b_cls)):
                    _mg.launch([_mg.Job(R * T, V, H, H2[1:], _mg.ax(RH, div=T, hi=H), _mg.ax(1),
                                        W_cls, _mg.ax(1), _mg.ax(H), logits, _mg.ax(V), bias=b_cls)])
            else:
                H2n = H2[1:].permute(1, 0, 2).contiguous()              # (R,T,H)
                logits = torch.addmm(b_cls, H2n.view(R * T, H), W_cls.t()).view(R, T, -1)
            attn = ALPHA.permute(1, 2, 0).contiguous()              # (R,K,T)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(*params)
            ctx.stash = (words, tf, O, M, wa, H1, H2, X1, X2, S1, S2, QL, ALPHA, ATT, H2n)
            ctx.dims = (R, K, F, E, H, T)
            ctx.coef = (COEF, S)
            ctx.words_need_grad = ctx.needs_input_grad[0]
        ctx.mark_non_differentiable(attn)
        ctx.set_materialize_grads(False)
        return logits, attn

    @staticmethod
    def backward(ctx, dlogits, _dattn):
        if dlogits is None:
            return (None,) * len(ctx.needs_input_grad)
        (W_td, b_td, W_ih1, W_hh1, b_ih1, b_hh1, W_f, W_h, w_a, W_lang, b_lang,
         W_ih2, W_hh2, b_ih2, b_hh2, W_cls, b_cls) = ctx.saved_tensors
        words, tf, O, M, wa, H1, H2, X1, X2, S1, S2, QL, ALPHA, ATT, H2n = ctx.stash
        R, K, F, E, H, T = ctx.dims
        dev = dlogits.device
        with torch.cuda.device(dev):
            dl = dlogits.contiguous().view(R * T, -1)
            V, RH = dl.shape[1], R * H
            pre_jobs = []
            if USE_MGEMM:
                dW_cls = torch.empty(V, H, device=dev)
                dH2 = torch.zeros(T, R, H, device=dev)          # k ranges add into it
                h2rows = _mg.ax(RH, div=T, hi=H)                # index (r, t) -> H2[t + 1, r] / dH2[t, r]
                pre_jobs = []
                # dW_cls = dl^T H2 (reduction over the R T rows) and dH2 = dl W_cls (reduction over V,
                # split over 16 workgroups per tile) on the matrix cores where the layout allows
                if not (MFMA_CLASSIFIER and _mg.mfma(_mg.TN, V, H, R * T, dl, _mg.ax(V), H2[1:], h2rows,
                                                     dW_cls, _mg.ax(H))):
                    pre_jobs.append(_mg.Job(V, H, R * T, dl, _mg.ax(1), _mg.ax(V), H2[1:], h2rows,
                                            _mg.ax(1), dW_cls, _mg.ax(H)))
                if not (MFMA_CLASSIFIER and _mg.mfma(_mg.NN, R * T, H, V, dl, _mg.ax(V), W_cls,
                                                     _mg.ax(H), dH2, h2rows, ksplit=16)):
                    pre_jobs.append(_mg.Job(R * T, H, V, dl, _mg.ax(V), _mg.ax(1), W_cls, _mg.ax(H),
                                            _mg.ax(1), dH2, h2rows, ksplit=16))
            else:
                if H2n is None:
                    H2n = H2[1:].permute(1, 0, 2).contiguous()
                dW_cls = torch.mm(dl.t(), H2n.view(R * T, H))
                dH2 = torch.mm(dl, W_cls).view(R, T, H).permute(1, 0, 2).contiguous()
            # transposed weights (the same small_linear kernel serves W^T products) and the
            # zeroed accumulators: one batched launch (s2c_batch_prep) instead of ten
            e = lambda *s: torch.empty(*s, device=dev)
            srcs = (W_ih2, W_hh2, W_lang, W_h, W_ih1, W_hh1, W_td[:, E:E + H])
            WTs = [e(w.shape[1], w.shape[0]) for w in srcs]
            dM, dwa_rows, dh1c = e(R, K, H), e(R, H), e(R, H)
            _batch_prep(srcs, WTs, (dM, dwa_rows, dh1c))
            WT_ih2, WT_hh2, WT_lang, WT_h, WT_ih1, WT_hh1, WT_td_h2 = WTs
            dh2_part = e(R, H)
            DA1 = e(T, R, E)
            # [dq | da2] side by side: one operand of the concatenated dh1 product below
            DQA = e(T, R, H + E)
            DQ, DA2 = DQA[:, :, :H], DQA[:, :, H:]
            DG = e(4, T, R, 3 * H)
            DGI1, DGH1, DGI2, DGH2 = DG[0], DG[1], DG[2], DG[3]
            COEF, S = ctx.coef
            persist = (COEF is not None and
                       _plib().s2c_decoder_bwd_persist_supported(R, K, H, E, T) == 1)
            fuse_bwd = persist or (FUSE_ATTN_X2_BWD and K <= ATTN_X2_MAX_K and 32 <= F <= 256
                                   and F & (F - 1) == 0 and E <= 512)
            # not fused: DV = [datt | dh1 via map_lang]; fused: datt never leaves the attention
            # backward and dh1 = [W_h^T | W_lang[:, F:]^T] [dq | da2] is ONE product
            DV = None if fuse_bwd else e(T, R, F + H)
            WT_hl = torch.cat([WT_h, WT_lang[F:]], 1) if fuse_bwd else None     # (H, H + E)
            DQc = None if fuse_bwd else e(T, R, H)
            dh2_direct, dh1_direct = e(R, H), e(R, H)
            if pre_jobs and not persist:
                _mg.launch(pre_jobs)
                pre_jobs = []
            if persist:
                # the whole recurrence in one kernel (csrc/s2c_decoder_persist.hip); the attention
                # backward's datt only enters through <da2, P_k> and <da2, Latt_t>
                if USE_MGEMM:
                    P, Latt = e(R, K, E), e(T, R, E)
                    pre_jobs += [_mg.mm(O.view(R * K, F), W_lang[:, :F].t(), P.view(R * K, E)),
                                 _mg.mm(ATT.view(T * R, F), W_lang[:, :F].t(), Latt.view(T * R, E))]
                    _mg.launch(pre_jobs)
                    pre_jobs = []
                else:
                    P = torch.matmul(O, W_lang[:, :F].t()).contiguous()          # (R,K,E)
                    Latt = torch.matmul(ATT, W_lang[:, :F].t()).contiguous()     # (T,R,E)
                xbuf, ctl = _persist_scratch(dev, H, E, bwd=True)
                a = _DecBwdArgs()
                a.R, a.K, a.H, a.E, a.T = R, K, H, E, T
                for n, v in (("dH2", dH2), ("C", COEF), ("S", S), ("X1", X1), ("X2", X2),
                             ("QL", QL), ("ALPHA", ALPHA), ("M", M), ("wa", wa), ("P", P),
                             ("Latt", Latt), ("WT_ih2", WT_ih2), ("WT_hh2", WT_hh2),
                             ("WT_hl", WT_hl), ("WT_ih1", WT_ih1), ("WT_hh1", WT_hh1),
                             ("WT_td", WT_td_h2), ("DA1", DA1), ("DQA", DQA), ("DG", DG),
                             ("dM", dM), ("dwa_rows", dwa_rows), ("xbuf", xbuf)):
                    assert v.is_contiguous(), n
                    setattr(a, n, v.data_ptr())
                a.prof = PROF_BWD.data_ptr() if PROF_BWD is not None else None
                a.nonce, a.started, a.fail = ctl.data_ptr(), ctl.data_ptr() + 4, ctl.data_ptr() + 8
                if _C.TIMER.enabled:     # transposed weights once + saved tensors in, gradients out
                    _C.TIMER.alg_bytes = 4 * (E * H + 6 * H * (E + H) + (H + E) * H +
                                              T * R * (10 * H + 4 * E + K) +
                                              T * R * (13 * H + 2 * E))
                _C.call("s2c_decoder_bwd_persist", ctypes.byref(a), _C.stream_ptr())
            # 6 launches per step.  GRU-2's gate gradients of step t-1 come out of the
            # epilogue of step t's last product (value = dh2 of step t-1); only the
            # very first needs its own launch.
            if not persist:
                _call("s2c_gru_gates_bwd", R, H, _p(dH2[T - 1]), None, _p(S2[0][T - 1]),
                      _p(S2[1][T - 1]), _p(S2[2][T - 1]), _p(S2[3][T - 1]), _p(H2[T - 1]),
                      _p(DGI2[T - 1]), _p(DGH2[T - 1]), _p(dh2_direct))
            for t in range(T - 1 if not persist else -1, -1, -1):
                _lin_pair(R,
                          _desc(E, 3 * H, WT_ih2, 3 * H, DGI2[t], 3 * H, DA2[t], H + E,
                                gate=X2[t], ldg=E, epi=2),
                          _desc(H, 3 * H, WT_hh2, 3 * H, DGH2[t], 3 * H, dh2_part, H,
                                add1=dh2_direct, ld1=H))
                if fuse_bwd:
                    _call("s2c_attn_bwd_x2", R, K, H, F, E, _p(DA2[t]), H + E, _p(WT_lang), E,
                          _p(ATT[t]), F, _p(ALPHA[t]), _p(O), _p(M), _p(QL[t]), H + E, _p(wa),
                          _p(dM), _p(DQ[t]), H + E, _p(dwa_rows),
                          alg_bytes=4 * (F * E + R * K * (3 * H + F) + R * (2 * H + 2 * K + F + E)))
                    _lin_pair(R, _desc(H, H + E, WT_hl, H + E, DQA[t], H + E, None, H,
                                       add1=dh1c, ld1=H),
                              gates=_gates(S1, t, H1[t], DGI1[t], DGH1[t], dh1_direct))
                else:
                    _lin_pair(R, _desc(F + H, E, WT_lang, E, DA2[t], H + E, DV[t], F + H))
                    _call("s2c_attn_bwd", R, K, H, F, _p(DV[t]), F + H, _p(ATT[t]), F,
                          _p(ALPHA[t]), _p(O), _p(M), _p(QL[t]), H + E, _p(wa), _p(dM),
                          _p(DQc[t]), _p(dwa_rows),
                          alg_bytes=4 * (R * K * (3 * H + F) + R * (2 * H + 2 * K + 2 * F)))
                    _lin_pair(R, _desc(H, H, WT_h, H, DQc[t], H, None, H, add1=DV[t][:, F:],
                                       ld1=F + H, add2=dh1c, ld2=H),
                              gates=_gates(S1, t, H1[t], DGI1[t], DGH1[t], dh1_direct))
                _lin_pair(R,
                          _desc(E, 3 * H, WT_ih1, 3 * H, DGI1[t], 3 * H, DA1[t], E,
                                gate=X1[t], ldg=E, epi=2),
                          _desc(H, 3 * H, WT_hh1, 3 * H, DGH1[t], 3 * H, dh1c, H,
                                add1=dh1_direct, ld1=H))
                if t > 0:
                    _lin_pair(R, _desc(H, E, WT_td_h2, E, DA1[t], E, None, H,
                                       add1=dh2_part, ld1=H, add2=dH2[t - 1], ld2=H),
                              gates=_gates(S2, t - 1, H2[t - 1], DGI2[t - 1],
                                           DGH2[t - 1], dh2_direct))
            # every bias gradient (and the sum over time of DA1) in ONE launch
            TR = T * R
            da1, da2 = DA1.view(TR, E), DQA.view(TR, H + E)[:, H:]
            dq_all = DQA.view(TR, H + E)[:, :H] if fuse_bwd else DQc.view(TR, H)
            gi1, gh1 = DGI1.view(TR, 3 * H), DGH1.view(TR, 3 * H)
            gi2, gh2 = DGI2.view(TR, 3 * H), DGH2.view(TR, 3 * H)
            (db_cls, DA1s, db_td, db_ih1, db_hh1, db_lang, db_ih2, db_hh2, dwa) = \
                fused.row_sums([dl, DA1.view(T, R * E), da1, gi1, gh1, da2, gi2, gh2, dwa_rows])
            DA1s = DA1s.view(R, E)
            if USE_MGEMM:
                # ---- stage 1: everything that depends on the recurrence only (one launch) -------
                dtf = e(R, F)
                dW_td, dW_lang = torch.empty_like(W_td), torch.empty_like(W_lang)
                dW_ih1, dW_hh1 = torch.empty_like(W_ih1), torch.empty_like(W_hh1)
                dW_ih2, dW_hh2 = torch.empty_like(W_ih2), torch.empty_like(W_hh2)
                dW_h, dW_f = torch.empty_like(W_h), torch.empty_like(W_f)
                dO = e(R, K, F)
                dMf = dM.view(R * K, H)
                ldtd, ldlang = W_td.stride(0), W_lang.stride(0)

                def wgrad(G, ldg, n_out, X, xrows, xcols, C, ldc_):
                    """C (n_out x xcols) = G^T X over the TR rows: G (TR x n_out, row stride ldg) in
                    (t, r) order, X rows addressed by `xrows` in the same order"""
                    return _mg.Job(n_out, xcols, TR, G, _mg.ax(1), _mg.ax(ldg), X, xrows, _mg.ax(1),
                                   C, _mg.ax(ldc_))
                tr_rows = lambda ld: _mg.ax(ld)                        # (T,R,.) tensors: row t R + r
                words_rows = _mg.ax(T * E, div=R, hi=E)                # words (R,T,E): index (t, r)
                jobs = [
                    # dtf = DA1s W_td[:, E+H:]
                    _mg.mm(DA1s, W_td[:, E + H:], dtf),
                    # dW_td column blocks: words | h2 (previous step) | target features
                    wgrad(da1, E, E, words, words_rows, E, dW_td, ldtd),
                    wgrad(da1, E, E, H2, tr_rows(H), H, dW_td[:, E:], ldtd),
                    _mg.mm(DA1s.t(), tf, dW_td[:, E + H:]),
                    wgrad(gi1, 3 * H, 3 * H, X1, tr_rows(E), E, dW_ih1, E),
                    wgrad(gh1, 3 * H, 3 * H, H1, tr_rows(H), H, dW_hh1, H),
                    wgrad(dq_all, dq_all.stride(0), H, H1[1:], tr_rows(H), H, dW_h, H),
                    wgrad(da2, da2.stride(0), E, ATT, tr_rows(F), F, dW_lang, ldlang),
                    wgrad(da2, da2.stride(0), E, H1[1:], tr_rows(H), H, dW_lang[:, F:], ldlang),
                    wgrad(gi2, 3 * H, 3 * H, X2, tr_rows(E), E, dW_ih2, E),
                    wgrad(gh2, 3 * H, 3 * H, H2, tr_rows(H), H, dW_hh2, H),
                    # dW_f = dM^T O ; the map_feat part of dO = dM W_f
                    _mg.mm(dMf.t(), O.view(R * K, F), dW_f),
                    _mg.mm(dMf, W_f, dO.view(R * K, F)),
                ]
                if fuse_bwd:
                    DATT = e(T, R, F)
                    jobs.append(_mg.mm(da2, W_lang[:, :F], DATT.view(TR, F)))
                else:
                    DATT = DV[:, :, :F]
                dwords = None
                if ctx.words_need_grad:
                    dwords = e(R, T, E)         # rows (r, t) <- DA1 (T,R,E) rows (t, r)
                    jobs.append(_mg.Job(R * T, E, E, DA1, _mg.ax(R * E, div=T, hi=E), _mg.ax(1),
                                        W_td, _mg.ax(ldtd), _mg.ax(1), dwords, _mg.ax(E)))
                _mg.launch(jobs)
                # ---- stage 2: dO += sum_t alpha_t (x) datt_t, per batch row (reads DATT) ---------
                ldatt_t = DATT.stride(0)
                _mg.launch([_mg.Job(K, F, T, ALPHA[:, r], _mg.ax(1), _mg.ax(R * K),
                                    DATT[:, r], _mg.ax(ldatt_t), _mg.ax(1), dO[r], _mg.ax(F),
                                    accumulate=True) for r in range(R)])
            else:
                # no recurrence through these two: hoisted out of the time loop
                dtf = torch.mm(DA1s, W_td[:, E + H:])                             # (R,F)
                # datt of every step (no recurrence through it): the not-fused path has it in DV
                DATT = (torch.matmul(da2, WT_lang[:F].t()).view(T, R, F) if fuse_bwd else DV[:, :, :F])
                dO = torch.bmm(ALPHA.permute(1, 2, 0), DATT.permute(1, 0, 2))
                # ---- every weight gradient: one stacked GEMM each ------------------
                # (column blocks written in place by the GEMMs: `out=` on a row-strided view is a
                # plain ldc for the library -- no temporaries, no copy kernels)
                dW_td = torch.empty_like(W_td)
                torch.mm(da1.t(), words.permute(1, 0, 2).reshape(TR, E), out=dW_td[:, :E])
                torch.mm(da1.t(), H2[:-1].reshape(TR, H), out=dW_td[:, E:E + H])
                torch.mm(DA1s.t(), tf, out=dW_td[:, E + H:])
                dW_ih1 = torch.mm(gi1.t(), X1.view(TR, E))
                dW_hh1 = torch.mm(gh1.t(), H1[:-1].reshape(TR, H))
                h1n = H1[1:].reshape(TR, H)
                dW_h = torch.mm(dq_all.t(), h1n)
                dW_lang = torch.empty_like(W_lang)
                torch.mm(da2.t(), ATT.view(TR, F), out=dW_lang[:, :F])
                torch.mm(da2.t(), h1n, out=dW_lang[:, F:])
                dW_ih2 = torch.mm(gi2.t(), X2.view(TR, E))
                dW_hh2 = torch.mm(gh2.t(), H2[:-1].reshape(TR, H))
                dMf = dM.view(R * K, H)
                dW_f = torch.mm(dMf.t(), O.view(R * K, F))
                dO = dO + torch.mm(dMf, W_f).view(R, K, F)
                dwords = None
                if ctx.words_need_grad:
                    dwords = torch.matmul(DA1.permute(1, 0, 2), W_td[:, :E])
        ctx.stash = None
        return (dwords, dtf, dO, None, None, dW_td, db_td, dW_ih1, dW_hh1, db_ih1,
                db_hh1, dW_f, dW_h, dwa.view_as(w_a), dW_lang, db_lang, dW_ih2, dW_hh2,
                db_ih2, db_hh2, dW_cls, db_cls)


def decode(module, word_embs, target_feats, obj_feats, masks, steps):
    """Run the fused decoder with the parameters of a TopDownSceneCaptionModule."""
    m = module
    params = (m.map_topdown[0].weight, m.map_topdown[0].bias,
              m.recurrent_cell_1.weight_ih, m.recurrent_cell_1.weight_hh,
              m.recurrent_cell_1.bias_ih, m.recurrent_cell_1.bias_hh,
              m.map_feat.weight, m.map_hidd.weight, m.attend.weight,
              m.map_lang[0].weight, m.map_lang[0].bias,
              m.recurrent_cell_2.weight_ih, m.recurrent_cell_2.weight_hh,
              m.recurrent_cell_2.bias_ih, m.recurrent_cell_2.bias_hh,
              m.classifier.weight, m.classifier.bias)
    return TopDownDecode.apply(word_embs, target_feats, obj_feats, masks, steps, *params)
