"""Point-major set-abstraction path (MI355X-first; kernels in csrc/s2c_sa.hip,
C ABI in include/s2c_fused.h).

`sa_group_mlp_pool` computes, for one set-abstraction stage, exactly what the
reference computes with QueryAndGroup -> SharedMLP -> max_pool2d
(pointnet2_modules.py:244-257), but on point-major rows: rows = (scene, centre,
sample), channels contiguous.  The 1x1 convolutions become row-major GEMMs
(the hand-written MFMA kernels of csrc/s2c_gemm*.hip, s2c_pgemm.hip, s2c_sgemm.hip); BatchNorm statistics,
BN+ReLU, BN+ReLU+max and all their backward passes are the hand-written
HBM-bound kernels.  `mlp_rows` is the same machinery without grouping/pooling
for the FP / voting / proposal heads.

Both are torch.autograd.Functions with hand-written backward, so nothing of
torch's conv / batch-norm / pooling machinery (MIOpen) is on the hot path.
"""
import ctypes
import os as _os

import torch
from torch.autograd import Function

from .. import _C

_I, _L, _P, _F = ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p, ctypes.c_float

_C.register("s2c_sa_gather_rows", [_I, _I, _I, _I, _I, _L, _L, _F, _I, _P, _P, _P, _P, _P, _P])
_C.register("s2c_sa_scatter_rows", [_I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P, _P])
_C.register("s2c_sa_scatter_sum", [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P])
_C.register("s2c_point_gemm", [_L, _I, _I, _P, _L, _P, _I, _P, _I, _P])
_C.register("s2c_point_gemm_stream", [_L, _I, _I, _P, _L, _P, _I, _P, _I, _P])
# tall per-point products (SA1 at the BASELINE sizes: 320 000 points x 132 channels) on the streaming
# kernel (bf16x3 split, LDS-DMA ring over the cloud's 540-byte rows read in place) instead of the exact
# fp32 MFMA chain of csrc/s2c_pgemm.hip, which is matrix-bound at 1/16 of the bf16 rate there (91 us
# where the 255 MB it moves allow ~55); the small stages and the fixtures (< 131072 points) stay on
# the exact chain.  = False: s2c_point_gemm everywhere.
POINT_GEMM_STREAM = _os.environ.get("S2C_POINT_GEMM_STREAM", "1") != "0"


def _point_gemm(P, f2, Wf, n_points, Cout, C, alg_bytes, alg_flops):
    """P (n_points x Cout) = f2 (n_points x C, any row stride) Wf^T."""
    if POINT_GEMM_STREAM and _gemm_split_on():
        if _C.TIMER.enabled:
            _C.TIMER.alg_bytes, _C.TIMER.alg_flops = int(alg_bytes), int(alg_flops)
            _C.TIMER.label = "s2c_sa_point_gemm"
        with torch.cuda.device(P.device):
            rc = _C.call("s2c_point_gemm_stream", n_points, Cout, C, f2.data_ptr(), f2.stride(0),
                         Wf.data_ptr(), Wf.stride(0), P.data_ptr(), Cout, _C.stream_ptr(), allow=(-2,))
        if rc == 0:
            return
    _call("s2c_point_gemm", P, n_points, Cout, C, f2.data_ptr(), f2.stride(0),
          Wf.data_ptr(), Wf.stride(0), P.data_ptr(), Cout,
          alg_bytes=alg_bytes, alg_flops=alg_flops, label="s2c_sa_point_gemm")
_C.register("s2c_sa_gather_add", [_I, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P])
_C.register("s2c_sa_scatter_sum_bn_bwd", [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P])
_C.register("s2c_fp_interp_rows", [_I, _I, _I, _I, _I, _P, _P, _P, _P, _L, _L, _P, _P])
_C.register("s2c_fp_interp_rows_grad", [_I, _I, _I, _I, _I, _P, _P, _P, _P, _P])
_C.register("s2c_rows_gemm_bn_eval", [_L, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _F, _I, _I, _P, _I, _P])
_C.register("s2c_sa_gather_gemm_bn_eval", [_I, _I, _I, _I, _I, _L, _L, _F, _I, _P, _P, _P, _P, _I, _P, _I,
                                           _P, _P, _P, _P, _F, _I, _I, _P, _I, _P])
_C.register("s2c_bn_train_stats", [_L, _I, _P, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P])
_C.register("s2c_bn_eval_coeffs", [_I, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P])
_C.register("s2c_bn_relu", [_L, _I, _P, _P, _P, _P, _I, _P])
_C.register("s2c_bn_relu_max", [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P])
_C.register("s2c_bn_relu_bwd", [_L, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P])
_C.register("s2c_rows_gemm", [_L, _I, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _P])
_C.register("s2c_sa_gather_gemm", [_I, _I, _I, _I, _I, _L, _L, _F, _I, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _P])
_C.register("s2c_rows_gemm_bn_relu_side", [_L, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P])
_C.register("s2c_bn_finalize_partials", [_I, _L, _I, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P])
_C.register("s2c_bn_finalize_partials_wt", [_I, _L, _I, _P, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                            _I, _I, _P, _P])
# the transposed weight the backward's tall input-gradient GEMMs read (Y = A Wt^T kernels) leaves with the
# forward's BatchNorm finalize launch of the same layer instead of a copy kernel per layer in the backward
WT_FROM_FINALIZE = True
_C.register("s2c_weight_grad", [_L, _I, _I, _P, _L, _P, _L, _P, _I, _P, _P, _P])
_C.register("s2c_bn_relu_max_bwd", [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P])
_C.register("s2c_bn_relu_bwd_stats", [_L, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P])
_C.register("s2c_bn_bwd_gemm", [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _I, _P])


def _ptr(t):
    return t.data_ptr() if t is not None else None


def _call(name, ref, *args, alg_bytes=0, alg_flops=0, label=None):
    if _C.TIMER.enabled:
        _C.TIMER.alg_bytes = int(alg_bytes)
        _C.TIMER.alg_flops = int(alg_flops)
        _C.TIMER.label = label
    with torch.cuda.device(ref.device):
        _C.call(name, *args, _C.stream_ptr())


def set_gemm_c64(on):
    """N > 64 problems on rows_gemm_c64_kernel (True, default) or the 32-k-slice kernel
    (csrc/s2c_gemm.hip); returns the previous setting."""
    lib = _C.load()
    lib.s2c_gemm_set_c64.argtypes = [_I]
    lib.s2c_gemm_set_c64.restype = _I
    return bool(lib.s2c_gemm_set_c64(int(bool(on))))


def _stat_blocks(M):
    lib = _C.load()
    lib.s2c_bn_stat_blocks.argtypes = [_L]
    lib.s2c_bn_stat_blocks.restype = _I
    return lib.s2c_bn_stat_blocks(M)


def _gemm_blocks(M, N):
    lib = _C.load()
    lib.s2c_rows_gemm_blocks.argtypes = [_L, _I]
    lib.s2c_rows_gemm_blocks.restype = _I
    return lib.s2c_rows_gemm_blocks(M, N)


# use the hand-written MFMA GEMM (statistics in its epilogue) for BN layers
USE_MFMA_GEMM = True
# weight gradient of a gather-fused first layer from point-indexed sums when its
# inputs need no gradient (no re-materialised operand).  (With input gradients -- SA2-SA4, where
# dY exists anyway -- the same sums were measured in round 3 and do not pay: 124 us of float
# atomics at SA2 against 91 us of re-materialising + the split-K product; step 9.86 vs 9.84 ms.)
SCATTER_DW = True
# the first layer's dY (it has no input gradient to feed) formed inside the point-sum kernel
# from (dA, Y) instead of written by the BN-backward pass and read back: = False = off
FUSE_DY_SCATTER = True


# The first layer of a gather stack in POINT SPACE (csrc/s2c_sa.hip: sa_gather_add): the feature
# product runs once per point (P = feats W_f^T), the gathered rows are P[idx] + W_x rel; the
# backward sums dY per point first (Z, S) and multiplies afterwards -- weight AND input gradients
# from B n point rows instead of B m ns gathered rows.  = False: the round-3 gather GEMM.
POINT_SPACE = True
POINT_SPACE_BWD = True
# The per-point product P runs on the exact fp32 matrix instruction (csrc/s2c_pgemm.hip: an fp32 FMA
# chain in k order, bit-identical to the tiled kernel's exact path), not on the bf16x3 split.  Both
# are fp32-accurate (rms error against float64 6e-7 vs 5e-7 of |P| ~ 2 on the golden model's own
# operands), but the train-mode gradients of the golden fixtures hang on discrete decisions behind the
# vote aggregation (ReLU masks / max aggregations a few ulps from a tie, each worth per cents of a small
# fixture's weight gradients: tools/diag_golden_ab.py): with P from the split kernel the c132 backbone
# gradients land 6.5e-2 of scale from the reference's, with another k order of the exact chain the cfg1
# ones 2.5e-2 -- with THIS chain, like the gather GEMM and the op-by-op path, both fixtures pass.


def _gather_add_blocks(M):
    lib = _C.load()
    lib.s2c_sa_gather_add_blocks.argtypes = [_L]
    lib.s2c_sa_gather_add_blocks.restype = _I
    return lib.s2c_sa_gather_add_blocks(M)


# Backward of a BN(+ReLU) layer: statistics pass, then ONE kernel that forms dY in the operand
# load of the input-gradient GEMM dX = dY W (s2c_bn_bwd_gemm) -- no apply pass, no library GEMM
# the BN+ReLU pass between two layers folded into the next layer's streaming GEMM (the
# activated operand leaves as a side output of csrc/s2c_gemm2.hip): = False = off
FUSE_BNRELU_GEMM = True
FUSE_BWD_GEMM = True
# forward of the layers that carry a bias (EdgeConv, the voting module's convs, the proposal head's
# last conv) on the hand GEMM instead of torch.addmm (= False: the library)
BIAS_LAYERS_BY_HAND = True


# Which hand-written kernel takes a backward product (measured on MI355X, tools/bench_bwd.py,
# tools/bench_dw_mid.py, tools/bench_da_mid.py, tools/bench_dwstream.py, tools/bench_sgemm.py); since
# round 5 the cfg3 train step makes no library GEMM call at all (tools/lib_gemm_census.py): tall weight
# gradients on csrc/s2c_dwstream.hip, small products on csrc/s2c_sgemm.hip.  torch.mm / bmm remain only
# as the fallback for layouts no kernel takes (non-unit column strides, exotic shapes).
def _fused_bwd_pays(M, C, N):
    return M >= 32768 and N % 4 == 0


def _hand_dw_pays(M, C, N):
    """dW = dY^T A (M rows, C x N output): the slab kernel of csrc/s2c_dw.hip + the partial-sum
    launch against the split-K library bmm + the same launch, measured on MI355X
    (tools/bench_dw_mid.py, us hand vs library): (16384,64,3) 23 vs 38 | (20480,128,128) 24 vs 38 |
    (20480,128,256) 25 vs 40 | (32768,128,128) 24 vs 39 | (32768,256,128) 32 vs 39 |
    (65536,128,128) 34 vs 37 | (32768,128,259) 52 vs 47 | (65536,256,128) 52 vs 47 |
    (262144,128,128) 101 vs 81 | (1M,64,64) 146 vs 120."""
    if M <= 16384:
        return True
    if N % 4 != 0:
        return False
    return M <= 32768 or (M <= 65536 and C * N <= 16384)


def _hand_da_pays(M, C, N):
    """dX (M x N) = dY (M x C) W: the streaming kernel for tall narrow layers, the 64-k-chunk
    kernel for N > 64 from 32768 rows on (tools/bench_da_mid.py, us library vs hand incl. the
    W^T copy: (32768,128,128) 21 vs 19 | (32768,128,259) 38 vs 33 | (65536,128,259) 65 vs 55 |
    (65536,256,128) 43 vs 40 | (262144,256,128) 149 vs 140 | (262144,128,131) 124 vs 127; below
    32768 rows hipBLASLt is ahead inside a replayed graph: (8192,256,256) 13 vs 20)."""
    return (M >= 262144 and N <= 64) or (M >= 32768 and N > 64)


def _gemm_split_on():
    lib = _C.load()
    lib.s2c_gemm_set_split.argtypes = [_I]
    lib.s2c_gemm_set_split.restype = _I
    on = lib.s2c_gemm_set_split(1)
    lib.s2c_gemm_set_split(on)
    return bool(on)


_C.register("s2c_small_gemm", [_L, _I, _I, _P, _L, _P, _L, _I, _P, _P, _L, _P])
# the small products of the layer stacks (2048 .. 32768 rows) on csrc/s2c_sgemm.hip instead of
# torch.mm / torch.addmm: dX = dY W reads W as stored (no transposed copy), y = x W^T + b
SMALL_GEMM = True


def _small_gemm_ok(M, N, K, lda, ldb, transposed):
    lib = _C.load()
    if not getattr(lib, "_sg_sized", False):
        lib.s2c_small_gemm_supported.restype = _I
        lib.s2c_small_gemm_supported.argtypes = [_L, _I, _I, _L, _L, _I]
        lib._sg_sized = True
    return bool(lib.s2c_small_gemm_supported(M, N, K, lda, ldb, int(transposed)))


def small_gemm(A, B, transposed, bias=None):
    """Y = A @ B (B: (K, N)) or A @ B^T (transposed: B is (N, K)) (+ bias) on s2c_small_gemm; None when
    the shape / layout is not taken (the caller falls back)."""
    if not (SMALL_GEMM and A.is_cuda and A.dtype == torch.float32 and B.dtype == torch.float32
            and A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1):
        return None
    M, K = A.shape
    N = B.shape[0] if transposed else B.shape[1]
    if (B.shape[1] if transposed else B.shape[0]) != K or M == 0:
        return None
    if not _small_gemm_ok(M, N, K, A.stride(0), B.stride(0), transposed):
        return None
    if A.data_ptr() % 4 or B.data_ptr() % (4 if transposed else 16):
        return None
    if bias is not None and not (bias.is_contiguous() and bias.dtype == torch.float32):
        return None
    Y = torch.empty((M, N), device=A.device)
    _call("s2c_small_gemm", Y, M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0),
          int(transposed), _ptr(bias), Y.data_ptr(), N,
          alg_bytes=4 * (M * K + K * N + M * N), alg_flops=2 * M * N * K)
    return Y


def _input_grad_gemm(dY, W, Wt=None):
    """dX (M, Cin) = dY (M, Cout) @ W (Cout, Cin) on the hand-written rows GEMM
    (Y = A Wt^T with Wt = W^T; Wt: the forward's copy, if it made one)."""
    M, Cout = dY.shape
    Cin = W.shape[1]
    if Wt is None:
        Wt = W.t().contiguous()
    dX = torch.empty((M, Cin), device=dY.device)
    _call("s2c_rows_gemm", dX, M, Cin, Cout, dY.data_ptr(), dY.stride(0), Wt.data_ptr(),
          Wt.stride(0), None, None, dX.data_ptr(), Cin, None,
          alg_bytes=4 * M * (Cout + Cin), alg_flops=2 * M * Cout * Cin)
    return dX


def set_gemm_split(on):
    """Products of the rows GEMMs: True = bf16x3 split on the bf16 matrix pipe (default,
    fp32-accurate), False = exact fp32 MFMA chain.  Returns the previous setting."""
    lib = _C.load()
    lib.s2c_gemm_set_split.argtypes = [_I]
    lib.s2c_gemm_set_split.restype = _I
    return bool(lib.s2c_gemm_set_split(int(bool(on))))


# Master switch of the point-major rows path (scan2cap_amd/opbyop.py flips it for A/B
# parity tests against the reference's op-by-op formulation).
ENABLED = True


def fused_available(t):
    return ENABLED and t.is_cuda


# ---------------------------------------------------------------------------
# gather rows
# ---------------------------------------------------------------------------
class _GatherRows(Function):
    """X (B*m*ns, 3+C) from xyz (B,N,3), new_xyz (B,m,3), point-major feats
    (B,N,C) (any row/batch stride, unit channel stride) and idx (B,m,ns)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats, idx, radius, normalize):
        B, N, _ = xyz.shape
        _, m, ns = idx.shape
        if feats is not None:
            if feats.stride(2) != 1:
                feats = feats.contiguous()
            C = feats.shape[2]
            frs, fbs = feats.stride(1), feats.stride(0)
        else:
            C, frs, fbs = 0, 0, 0
        xyz_c, new_c = xyz.contiguous(), new_xyz.contiguous()
        X = torch.empty((B * m * ns, 3 + C), dtype=torch.float32, device=xyz.device)
        _call("s2c_sa_gather_rows", xyz, B, N, m, ns, C, frs, fbs, float(radius),
              int(bool(normalize)), xyz_c.data_ptr(), new_c.data_ptr(), _ptr(feats),
              idx.data_ptr(), X.data_ptr(),
              alg_bytes=4 * (min(B * N, B * m * ns) * (3 + C) + B * m * ns
                             + B * m * ns * (3 + C)))
        ctx.save_for_backward(idx)
        ctx.dims = (B, N, m, ns, C, float(radius), int(bool(normalize)))
        ctx.need = (xyz.requires_grad or new_xyz.requires_grad,
                    feats is not None and feats.requires_grad)
        return X

    @staticmethod
    def backward(ctx, dX):
        (idx,) = ctx.saved_tensors
        B, N, m, ns, C, radius, normalize = ctx.dims
        need_xyz, need_feats = ctx.need
        dX = dX.contiguous()
        d_feats = torch.empty((B, N, C), dtype=torch.float32, device=dX.device) \
            if need_feats else None
        d_xyz = torch.empty((B, N, 3), dtype=torch.float32, device=dX.device) \
            if need_xyz else None
        d_new = torch.empty((B, m, 3), dtype=torch.float32, device=dX.device) \
            if need_xyz else None
        if need_xyz or need_feats:
            _call("s2c_sa_scatter_rows", dX, B, N, m, ns, C, radius, normalize,
                  dX.data_ptr(), idx.data_ptr(), _ptr(d_feats), _ptr(d_xyz),
                  _ptr(d_new),
                  alg_bytes=4 * (B * m * ns * (3 + C + 1) + B * N * C))
        return d_xyz, d_new, d_feats, None, None, None


# ---------------------------------------------------------------------------
# MLP over rows
# ---------------------------------------------------------------------------
class _FPRows(Function):
    """known (B,m,C2) point-major, skip (B,n,C1) point-major view or None, idx / weight
    (B,n,3) -> rows (B*n, C2+C1) = [three_interpolate | skip] (pointnet2_modules.py:398-410)."""

    @staticmethod
    def forward(ctx, known, skip, idx, weight):
        B, m, C2 = known.shape
        n = idx.shape[1]
        known = known.contiguous()
        if skip is not None and skip.stride(2) != 1:
            skip = skip.contiguous()
        C1 = skip.shape[2] if skip is not None else 0
        idx, weight = idx.contiguous(), weight.contiguous()
        out = torch.empty((B * n, C2 + C1), device=known.device)
        _call("s2c_fp_interp_rows", out, B, n, m, C2, C1, known.data_ptr(), idx.data_ptr(),
              weight.data_ptr(), _ptr(skip), skip.stride(1) if skip is not None else 0,
              skip.stride(0) if skip is not None else 0, out.data_ptr(),
              alg_bytes=4 * (B * n * (3 * C2 + C1 + 6) + B * n * (C2 + C1)))
        ctx.save_for_backward(idx, weight)
        ctx.dims = (B, n, m, C2, C1)
        return out

    @staticmethod
    def backward(ctx, dOut):
        idx, weight = ctx.saved_tensors
        B, n, m, C2, C1 = ctx.dims
        dOut = dOut.contiguous()
        d_known = None
        if ctx.needs_input_grad[0]:
            d_known = torch.empty((B, m, C2), device=dOut.device)
            _call("s2c_fp_interp_rows_grad", dOut, B, n, m, C2, C2 + C1, dOut.data_ptr(),
                  idx.data_ptr(), weight.data_ptr(), d_known.data_ptr(),
                  alg_bytes=4 * (B * n * (C2 + 6) + 3 * B * n * C2))
        d_skip = None
        if C1 > 0 and ctx.needs_input_grad[1]:
            d_skip = dOut.view(B, n, C2 + C1)[:, :, C2:]
        return d_known, d_skip, None, None


def fp_rows(known_pm, skip_pm, idx, weight):
    return _FPRows.apply(known_pm, skip_pm, idx, weight)


class LayerSpec(object):
    """One shared-MLP layer: Y = X W^T (+ bias) -> [BatchNorm] -> [ReLU]."""
    __slots__ = ("has_bias", "bn", "relu")

    def __init__(self, has_bias, bn, relu):
        self.has_bias, self.bn, self.relu = has_bias, bn, relu


_C.register("s2c_rows_gemm_pool_raw", [_L, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _P, _P, _P, _P, _I, _P, _P])
_C.register("s2c_pool_select", [_L, _I, _P, _P, _P, _P, _P])
_C.register("s2c_bn_bwd_gemm_next_stats", [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _I,
                                           _P, _P, _P, _P, _P, _I, _P, _P])
_C.register("s2c_bn_bwd_finalize_partials", [_I, _L, _I, _P, _I, _P, _P, _P, _P, _P, _P])
_C.register("s2c_bn_bwd_dx_dw64", [_L, _P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I,
                                   _P, _P, _P])
# a 64 -> 64 BatchNorm layer between two others (SA1's second layer): dY never written, dX, dW and the
# previous layer's column sums in one pass over (dA, Y, the previous layer's Y) -- csrc/s2c_bnbwd_fused.hip;
# = False: s2c_bn_bwd_gemm_next_stats + the dY tensor + s2c_weight_grad_stream
FUSE_BWD_DX_DW = True


def _bwd_dx_dw_parts(M):
    lib = _C.load()
    if not getattr(lib, "_fb_sized", False):
        lib.s2c_bn_bwd_dx_dw64_parts.restype = _I
        lib.s2c_bn_bwd_dx_dw64_parts.argtypes = [_L]
        lib._fb_sized = True
    return lib.s2c_bn_bwd_dx_dw64_parts(M)
_C.register("s2c_pool_bwd_input_grad_next_stats", [_L, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P, _P, _I,
                                                   _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P])
_C.register("s2c_rows_gemm_next_stats", [_L, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P])
_C.register("s2c_bn_relu_bwd_apply", [_L, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P])
# the column sums of a layer's BatchNorm backward out of the epilogue of the GEMM that produces
# its upstream gradient (s2c_bn_bwd_gemm_next_stats) instead of a statistics pass over (dA, Y);
# = False: the separate pass
BWD_STATS_IN_GEMM = True
# the extremum a pooled BatchNorm + ReLU layer will select out of the GEMM's epilogue also when Y
# is materialised (wide layers, 64-k-chunk kernel); = False: s2c_bn_relu_max over Y
POOL_EXT_IN_GEMM = True
_C.register("s2c_bn_relu_max_bwd_stats", [_L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P])
_C.register("s2c_pool_bwd_dk", [_L, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P])
_C.register("s2c_pool_bwd_sp", [_L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P])
_C.register("s2c_pool_bwd_prep", [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P])
_C.register("s2c_pool_bwd_final", [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P])
_C.register("s2c_pool_bwd_input_grad", [_L, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P, _P, _I, _P, _P, _I,
                                        _P])

# The pooled LAST layer of a training stack without its (M x C3) pre-activation / gradient
# tensors (DESIGN 4.3 "pooled layer algebra"): forward = raw extrema out of the GEMM epilogue,
# backward = input / weight gradients from the layer's INPUT activation.  = False: off.
POOL_ALGEBRA = True


def _pool_bwd_takes(M, N, KA, C3):
    lib = _C.load()
    if not hasattr(lib, "_s2c_pool_bwd_sig"):
        lib.s2c_pool_bwd_supported.argtypes = [_L, _I, _I, _I]
        lib.s2c_pool_bwd_supported.restype = _I
        lib._s2c_pool_bwd_sig = True
    return bool(lib.s2c_pool_bwd_supported(M, N, KA, C3))


def pool_algebra_takes(M, Cout, K_in, pool_ns):
    return (POOL_ALGEBRA and pool_ns in (16, 32, 64) and M % pool_ns == 0 and Cout <= 128
            and Cout % 8 == 0 and K_in in (32, 64, 128) and Cout <= (256 // K_in) * 32
            and pool_ns * K_in * 4 <= 16384 and (pool_ns * K_in * 4) % 4096 == 0   # s2c_pool_bwd_sp tiles
            and _gemm_split_on()
            and _stream_takes(M, Cout, K_in) and _pool_bwd_takes(M, K_in, K_in, Cout))


def pooled_layer_backward(dOut, arg, ymax, scale, shift, mean, invstd, gamma, frozen, A, W, ns,
                          need_dA=True, next_bn=None, act=None):
    """Gradients of out = max over ns rows of relu(BN(A W^T)) w.r.t. A (M x K), W (C3 x K),
    gamma, beta, given dOut (J x C3) -- without Y3 = A W^T or dY3 (M x C3 each):
        dY3 = dkrow - g (.) Y3 + e   per channel, g = k0 k2 invstd, e = g mean - k0 k1,
        dA  = dkrow W - A (W^T diag(g) W) + e W,
        dW  = SP - diag(g) W (A^T A) + e (x) colsum(A),   SP[c] = sum_j dk[j,c] A[row(j,c)].
    (coef k0 k1 k2 = the statistics of s2c_bn_relu_max_bwd; float64 for the small matrices.)
    act = (pscale, pshift, prelu): `A` is the PREVIOUS layer's pre-activation and the layer's input
    relu?(A pscale + pshift) is formed inside the three kernels that read it (the forward kept no copy)."""
    dev = dOut.device
    asc, ash, arelu = (act[0].data_ptr(), act[1].data_ptr(), int(act[2])) if act is not None \
        else (None, None, 0)
    J, C3 = dOut.shape
    M, K = A.shape
    nb = _stat_blocks(J)
    partial = torch.empty(nb * 2 * max(C3, 256), device=dev)
    coef = torch.empty(3 * C3, device=dev)
    has_affine = gamma is not None
    dgamma = torch.empty(C3, device=dev) if has_affine else None
    dbeta = torch.empty(C3, device=dev) if has_affine else None
    dOut = dOut.contiguous()
    _call("s2c_bn_relu_max_bwd_stats", dOut, J, ns, C3, dOut.data_ptr(), ymax.data_ptr(),
          scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), _ptr(gamma),
          int(frozen), partial.data_ptr(), coef.data_ptr(), _ptr(dgamma), _ptr(dbeta),
          alg_bytes=4 * 2 * J * C3)
    dk = torch.empty_like(dOut)
    arg16 = torch.empty((J, C3), dtype=torch.int16, device=dev)
    _call("s2c_pool_bwd_dk", dOut, J, C3, dOut.data_ptr(), ymax.data_ptr(), scale.data_ptr(),
          shift.data_ptr(), coef.data_ptr(), arg.data_ptr(), dk.data_ptr(), arg16.data_ptr(),
          alg_bytes=4 * 4 * J * C3)
    W = W.contiguous()
    # ---- input gradient --------------------------------------------------------------------
    dA = None
    if need_dA:
        Wcat = torch.empty((K, K + C3), device=dev)
        cvec = torch.empty(K, device=dev)
        ge = torch.empty(2 * C3, device=dev)
        _call("s2c_pool_bwd_prep", W, C3, K, coef.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
              W.data_ptr(), Wcat.data_ptr(), cvec.data_ptr(), ge.data_ptr())
        dA = torch.empty((M, K), device=dev)
        if next_bn is not None:
            # next_bn = the record of the layer whose upstream gradient this dA is: its
            # BatchNorm-backward column sums leave with the GEMM (next_bn["prestats"])
            nbg = _gemm_blocks(M, K)
            npart = torch.empty(nbg * 2 * K, device=dev)
            _call("s2c_pool_bwd_input_grad_next_stats", A, M, K, K, C3, ns, A.data_ptr(),
                  A.stride(0), arg16.data_ptr(), dk.data_ptr(), Wcat.data_ptr(), Wcat.stride(0),
                  cvec.data_ptr(), dA.data_ptr(), K, next_bn["Y"].data_ptr(),
                  next_bn["scale"].data_ptr(), next_bn["shift"].data_ptr(),
                  next_bn["mean"].data_ptr(), next_bn["invstd"].data_ptr(),
                  int(next_bn["relu"]), npart.data_ptr(), asc, ash, arelu,
                  alg_bytes=4 * (3 * M * K + 2 * J * C3), alg_flops=2 * M * K * (K + C3))
            next_bn["prestats"] = (npart, nbg)
        else:
            _call("s2c_pool_bwd_input_grad", A, M, K, K, C3, ns, A.data_ptr(), A.stride(0),
                  arg16.data_ptr(), dk.data_ptr(), Wcat.data_ptr(), Wcat.stride(0),
                  cvec.data_ptr(), dA.data_ptr(), K, asc, ash, arelu,
                  alg_bytes=4 * (2 * M * K + 2 * J * C3),
                  alg_flops=2 * M * K * (K + C3))
    # ---- weight gradient -------------------------------------------------------------------
    lib = _C.load()
    lib.s2c_pool_bwd_sp_blocks.argtypes = [_L]
    lib.s2c_pool_bwd_sp_blocks.restype = _I
    nblk = lib.s2c_pool_bwd_sp_blocks(J)
    sp = torch.empty((nblk, C3 * K + K), device=dev)
    _call("s2c_pool_bwd_sp", A, J, ns, C3, K, A.data_ptr(), arg.data_ptr(), dk.data_ptr(),
          sp.data_ptr(), asc, ash, arelu, alg_bytes=4 * (M * K + 2 * J * C3 + nblk * (C3 + 1) * K))
    # both partial tables are summed by ONE multi_colsum launch (kernel-boundary reduction).
    # NOT torch.sum: its cross-block reductions reset a semaphore buffer with hipMemsetAsync,
    # and memset nodes of a captured hipGraph are not ordered against kernels on ROCm 7.2
    # (DESIGN 5): the second replay of a two-graph step returned a wrong dW here.
    pending = []
    gram = _weight_grad_stream(A, A, pending, act=act) if act is not None else None
    if gram is None:
        if act is not None:
            raise _C.S2CError("pooled_layer_backward: the Gram matrix of a recomputed input needs the "
                              "streaming kernel (DW_STREAM)")
        gram = _weight_grad(A, A, pending)                  # A^T A  (K x K)
    spsum = torch.empty(C3 * K + K, device=dev)
    pending.append((sp.view(nblk, 1, C3 * K + K), spsum))
    flush_partial_sums(pending)
    dW = torch.empty((C3, K), device=dev)
    _call("s2c_pool_bwd_final", dW, C3, K, spsum.data_ptr(), gram.data_ptr(), W.data_ptr(),
          coef.data_ptr(), mean.data_ptr(), invstd.data_ptr(), dW.data_ptr())
    return dA, dW, dgamma, dbeta


def _next_takes_prologue(specs, params, pi, li, M, K):
    """Can layer li+1 (its parameters start at params[pi]) run as the streaming GEMM with the
    BN+ReLU prologue?  Same conditions as `gemm_stats` of the forward loop + the kernel's."""
    nsp = specs[li + 1]
    W = params[pi]
    bn = nsp.bn
    if not (USE_MFMA_GEMM and _gemm_split_on() and bn is not None and not nsp.has_bias
            and (bn.training or bn.running_mean is None) and W.stride(1) == 1
            and W.dtype == torch.float32 and W.is_cuda):
        return False
    if nsp is not None and specs[li].relu:
        # ReLU prologue: also the 64-k-chunk kernel (N > 64 layers the streaming kernel leaves)
        lib = _C.load()
        if not hasattr(lib, "_s2c_side_sig"):
            lib.s2c_rows_gemm_side_supported.argtypes = [_L, _I, _I]
            lib.s2c_rows_gemm_side_supported.restype = _I
            lib._s2c_side_sig = True
        return bool(lib.s2c_rows_gemm_side_supported(M, W.shape[0], K))
    return _stream_takes(M, W.shape[0], K)


def _stream_takes(M, N, K):
    lib = _C.load()
    if not hasattr(lib, "_s2c_stream_sig"):
        lib.s2c_rows_stream_supported.argtypes = [_L, _I, _I, _I]
        lib.s2c_rows_stream_supported.restype = _I
        lib._s2c_stream_sig = True
    return bool(lib.s2c_rows_stream_supported(M, N, K, 0))


def _activation_of(rec):
    """relu?(Y scale + shift) of a BatchNorm layer's record (s2c_bn_relu): the input of the next layer
    when the forward did not keep it (act_from_prev)."""
    Y = rec["Y"]
    A = torch.empty_like(Y)
    _call("s2c_bn_relu", Y, Y.shape[0], Y.shape[1], Y.data_ptr(), rec["scale"].data_ptr(),
          rec["shift"].data_ptr(), A.data_ptr(), int(rec["relu"]), alg_bytes=8 * Y.numel())
    return A


class _MLPRows(Function):
    """forward(X, specs, pool_ns, *params) with params = per layer
    [W (Cout,Cin), bias?, gamma?, beta?].  pool_ns > 0: the last layer's
    BN+ReLU is fused with a max over groups of pool_ns consecutive rows."""

    @staticmethod
    def forward(ctx, X, xyz, new_xyz, feats, gcfg, specs, pool_ns, *params):
        # Operand: either X (M, Cin), or -- X is None -- the gathered rows described
        # by (xyz, new_xyz, feats, gcfg=(idx, radius, normalize)); the first layer
        # then gathers on the fly (s2c_sa_gather_gemm) and X is never built.
        gather = None
        if X is None:
            gather = GatherSpec(xyz, new_xyz, feats, gcfg[0], gcfg[1], gcfg[2])
            gather.needs_grad = any(ctx.needs_input_grad[1:4])
            gather.need_xyz = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
            gather.need_feats = bool(ctx.needs_input_grad[3])
            dev, M = gather.xyz.device, gather.rows
        else:
            dev, M = X.device, X.shape[0]
        need_grad = any(ctx.needs_input_grad)
        saved = []          # per layer dict of tensors needed in backward
        A = X
        pi = 0
        nl = len(specs)
        partial = None
        out = None
        deferred = None     # (Y, scale, shift, relu) of the previous layer: BN+ReLU not applied yet
        pooled_raw = None   # raw per-centre extrema of the last layer (pool algebra path)
        pooled_ext = None   # the same out of the 64-k-chunk kernel, Y materialised
        for li, sp in enumerate(specs):
            W = params[pi]; pi += 1
            bias = None
            if sp.has_bias:
                bias = params[pi]; pi += 1
            gamma = beta = None
            bn = sp.bn
            if bn is not None:
                gamma, beta = params[pi], params[pi + 1]; pi += 2
            Cout = W.shape[0]
            train_stats = bn is not None and (bn.training or bn.running_mean is None)
            if train_stats and bn.momentum is None and bn.running_mean is not None:
                raise NotImplementedError(
                    "BatchNorm(momentum=None) (cumulative moving average) is not "
                    "implemented on the rows path; use a float momentum")
            from_gather = gather is not None and li == 0
            act_from_prev = False
            # (A bias in front of a train-mode BatchNorm cancels in the normalised output; running the
            # GEMM without it -- statistics out of the epilogue, momentum * bias added back to the
            # running mean -- was built and withdrawn: time-neutral, and the differently rounded
            # pre-activations flip a ReLU mask of the voting module against float64,
            # tests/test_modules_cfg3_gpu.py.)
            gemm_stats = (USE_MFMA_GEMM and train_stats and bias is None
                          and W.stride(1) == 1
                          and (from_gather or deferred is not None or A.stride(1) == 1))
            pre_activated = False
            point_space = (from_gather and POINT_SPACE and Cout % 4 == 0 and Cout <= 256
                           and bias is None and W.stride(1) == 1 and gather.feats2d() is not False)
            if point_space:
                g = gather
                P = None
                if g.C > 0:
                    f2, Wf = g.feats2d(), W[:, 3:]
                    P = torch.empty((g.B * g.N, Cout), device=dev)
                    # the op's contract (unique source rows + idx + Y) is split over the two launches;
                    # P is an intermediate, not algorithmic traffic
                    pb, pf = 4 * min(g.B * g.N, M) * g.C, 2 * g.B * g.N * g.C * Cout
                    _point_gemm(P, f2, Wf, g.B * g.N, Cout, g.C, pb, pf)
                nbg = _gather_add_blocks(M)
                gpart = torch.empty(nbg * 2 * Cout, device=dev) if gemm_stats else None
                Y = torch.empty((M, Cout), device=dev)
                _call("s2c_sa_gather_add", Y, g.B, g.N, g.m, g.ns, Cout, g.radius, g.normalize,
                      g.xyz.data_ptr(), g.new_xyz.data_ptr(), _ptr(P), g.idx.data_ptr(),
                      W.data_ptr(), W.stride(0), Y.data_ptr(), _ptr(gpart),
                      alg_bytes=4 * (min(g.B * g.N, M) * 3 + M + M * Cout))
            elif from_gather:
                assert bias is None and W.stride(1) == 1
                g = gather
                nbg = _gemm_blocks(M, Cout)
                gpart = torch.empty(nbg * 2 * Cout, device=dev) if gemm_stats else None
                Y = torch.empty((M, Cout), device=dev)
                _call("s2c_sa_gather_gemm", Y, g.B, g.N, g.m, g.ns, g.C, g.frs, g.fbs,
                      g.radius, g.normalize, g.xyz.data_ptr(), g.new_xyz.data_ptr(),
                      _ptr(g.feats), g.idx.data_ptr(), Cout, W.data_ptr(), W.stride(0),
                      Y.data_ptr(), Cout, _ptr(gpart),
                      alg_bytes=4 * (min(g.B * g.N, M) * (3 + g.C) + M + M * Cout),
                      alg_flops=2 * M * (3 + g.C) * Cout)
            elif (li == nl - 1 and pool_ns > 0 and gemm_stats and need_grad
                  and pool_algebra_takes(M, Cout, (deferred[0] if deferred is not None else A).shape[1],
                                         pool_ns)):
                # pooled last layer: raw per-centre extrema out of the GEMM's epilogue, the
                # (M x Cout) pre-activation is never written (pooled_layer_backward needs none)
                nbg = _gemm_blocks(M, Cout)
                gpart = torch.empty(nbg * 2 * Cout, device=dev)
                J = M // pool_ns
                # the extremum BN + ReLU + max will select (by the sign of gamma) and its row
                raws = (torch.empty((J, Cout), device=dev), torch.empty((J, Cout), dtype=torch.int32, device=dev))
                if deferred is not None:
                    pY, pscale, pshift, prelu = deferred
                    deferred = None
                    src, K_in = pY, pY.shape[1]
                    # the pooled-layer algebra reads the layer's input in three kernels; each forms it
                    # from pY itself (POOL_ALGEBRA_ACT): no 268 MB activation side output at SA1
                    if (POOL_ALGEBRA_ACT and DW_STREAM and pY.is_contiguous() and K_in <= 64
                            and pool_ns >= 32 and M >= DW_STREAM_MIN_ROWS
                            and _dw_stream_parts(M, K_in, K_in, pY, pY) > 0):
                        act_from_prev = "algebra"
                    A = None if act_from_prev else torch.empty_like(pY)
                    pro = (pscale.data_ptr(), pshift.data_ptr(), int(prelu), _ptr(A), K_in)
                else:
                    src, K_in = A, A.shape[1]
                    pro = (None, None, 0, None, 0)
                _call("s2c_rows_gemm_pool_raw", src, M, Cout, K_in, src.data_ptr(), src.stride(0),
                      pro[0], pro[1], pro[2], pro[3], pro[4], W.data_ptr(), W.stride(0), pool_ns,
                      _ptr(gamma),
                      raws[0].data_ptr(), raws[1].data_ptr(), None, 0, gpart.data_ptr(),
                      alg_bytes=4 * (M * K_in * (2 if pro[0] else 1) + 2 * J * Cout),
                      alg_flops=2 * M * K_in * Cout)
                Y = None
                pooled_raw = raws
            elif (POOL_EXT_IN_GEMM and li == nl - 1 and pool_ns in (16, 32, 64) and gemm_stats
                  and M % pool_ns == 0 and Cout > 64 and _gemm_split_on()
                  and (deferred is not None or (A.stride(1) == 1 and A.dtype == torch.float32))
                  and (deferred is None or deferred[3])):
                # pooled last layer, Y materialised (the backward reads it): the extremum that BN +
                # ReLU + max will select leaves with the GEMM (64-k-chunk kernel), the pooled pass
                # over Y (s2c_bn_relu_max) shrinks to s2c_pool_select on J x Cout values
                nbg = _gemm_blocks(M, Cout)
                gpart = torch.empty(nbg * 2 * Cout, device=dev)
                Y = torch.empty((M, Cout), device=dev)
                J = M // pool_ns
                raws = (torch.empty((J, Cout), device=dev),
                        torch.empty((J, Cout), dtype=torch.int32, device=dev))
                if deferred is not None:
                    pY, pscale, pshift, prelu = deferred
                    deferred = None
                    src, K_in = pY, pY.shape[1]
                    if need_grad and pY.is_contiguous() and _dw_will_stream(M, Cout, K_in, pY):
                        act_from_prev = "stream"   # no activation side output (DW_STREAM_ACT)
                    A = None if act_from_prev else torch.empty_like(pY)
                    pro = (pscale.data_ptr(), pshift.data_ptr(), int(prelu), _ptr(A), K_in)
                else:
                    src, K_in = A, A.shape[1]
                    pro = (None, None, 0, None, 0)
                rc = _C.call("s2c_rows_gemm_pool_raw", M, Cout, K_in, src.data_ptr(), src.stride(0),
                             pro[0], pro[1], pro[2], pro[3], pro[4], W.data_ptr(), W.stride(0),
                             pool_ns, _ptr(gamma), raws[0].data_ptr(), raws[1].data_ptr(),
                             Y.data_ptr(), Cout, gpart.data_ptr(), _C.stream_ptr(), allow=(-2,))
                if rc == 0:
                    pooled_ext = raws
                elif pro[0] is not None:
                    # not taken: the two-launch form of the same layer
                    _call("s2c_rows_gemm_bn_relu_side", Y, M, Cout, K_in, src.data_ptr(),
                          src.stride(0), pro[0], pro[1], pro[2], pro[3], pro[4], W.data_ptr(),
                          W.stride(0), Y.data_ptr(), Cout, gpart.data_ptr())
                else:
                    _call("s2c_rows_gemm", Y, M, Cout, K_in, src.data_ptr(), src.stride(0),
                          W.data_ptr(), W.stride(0), None, None, Y.data_ptr(), Cout,
                          gpart.data_ptr())
            elif deferred is not None:
                # the previous layer's BN+ReLU happens in THIS layer's operand load; the
                # activation (needed for the weight gradient) leaves as a side output
                pY, pscale, pshift, prelu = deferred
                deferred = None
                nbg = _gemm_blocks(M, Cout)
                gpart = torch.empty(nbg * 2 * Cout, device=dev)
                Y = torch.empty((M, Cout), device=dev)
                K_in = pY.shape[1]
                # the activation is kept for this layer's weight gradient -- unless the backward
                # recomputes it from pY inside its one pass (s2c_bn_bwd_dx_dw64): no side output
                act_from_prev = bool(
                    need_grad and FUSE_BWD_DX_DW and FUSE_BWD_GEMM and BWD_STATS_IN_GEMM
                    and bn is not None and bias is None and Cout == 64 and K_in == 64
                    and not (li == nl - 1 and pool_ns > 0) and pY.is_contiguous()
                    and _fused_bwd_pays(M, Cout, K_in) and M >= DW_STREAM_MIN_ROWS
                    and W.stride(1) == 1 and _gemm_split_on() and _bwd_dx_dw_parts(M) > 0)
                if (not act_from_prev and need_grad and bn is not None and pY.is_contiguous()
                        and _dw_will_stream(M, Cout, K_in, pY)):
                    act_from_prev = "stream"       # ... or the streaming weight gradient does
                A = None if act_from_prev else torch.empty_like(pY)
                _call("s2c_rows_gemm_bn_relu_side", Y, M, Cout, K_in, pY.data_ptr(), pY.stride(0),
                      pscale.data_ptr(), pshift.data_ptr(), int(prelu), _ptr(A), K_in,
                      W.data_ptr(), W.stride(0), Y.data_ptr(), Cout, gpart.data_ptr(),
                      alg_bytes=4 * ((1 if act_from_prev else 2) * M * K_in + M * Cout),
                      alg_flops=2 * M * K_in * Cout)
            elif gemm_stats:
                # hand-written f32 MFMA GEMM; BN batch statistics come out of its
                # epilogue as per-row-block partials (no extra pass over Y)
                nbg = _gemm_blocks(M, Cout)
                gpart = torch.empty(nbg * 2 * Cout, device=dev)
                Y = torch.empty((M, Cout), device=dev)
                K_in = A.shape[1]
                _call("s2c_rows_gemm", Y, M, Cout, K_in, A.data_ptr(), A.stride(0),
                      W.data_ptr(), W.stride(0), None, None, Y.data_ptr(), Cout,
                      gpart.data_ptr(), alg_bytes=4 * (M * K_in + M * Cout),
                      alg_flops=2 * M * K_in * Cout)
            elif (BIAS_LAYERS_BY_HAND and USE_MFMA_GEMM and bn is None and A.is_cuda
                  and A.dtype == torch.float32 and W.stride(1) == 1 and _gemm_split_on()):
                # BatchNorm-free layer (EdgeConv's linear layers, vgen.conv3, the proposal head's
                # last conv): bias (+ ReLU) in the hand GEMM's affine epilogue with identity scale;
                # what is kept for the backward's ReLU mask is the activation (a > 0 <=> y > 0)
                if A.stride(1) != 1:
                    A = A.contiguous()
                g_, b_, m_, v_, eps_ = _eval_affine(sp, bias, None, None, Cout, dev)
                K_in = A.shape[1]
                Y = torch.empty((M, Cout), device=dev)
                _call("s2c_rows_gemm_bn_eval", Y, M, Cout, K_in, A.data_ptr(), A.stride(0),
                      W.data_ptr(), W.stride(0), _ptr(g_), _ptr(b_), m_.data_ptr(), v_.data_ptr(),
                      eps_, int(sp.relu), 0, Y.data_ptr(), Cout,
                      alg_bytes=4 * (M * K_in + M * Cout), alg_flops=2 * M * K_in * Cout)
                pre_activated = True
            else:
                Y = small_gemm(A, W, True, bias)
                if Y is None:
                    Y = torch.addmm(bias, A, W.t()) if bias is not None else torch.mm(A, W.t())
            rec = {"A_in": None if from_gather else A, "W": W,
                   "has_bias": bias is not None, "point_space": point_space,
                   "act_from_prev": act_from_prev}
            last = li == nl - 1
            if bn is not None:
                scale = torch.empty(Cout, device=dev)
                shift = torch.empty(Cout, device=dev)
                mean = torch.empty(Cout, device=dev)
                invstd = torch.empty(Cout, device=dev)
                if gemm_stats and gpart is not None:
                    mom = bn.momentum if bn.momentum is not None else 0.0
                    if (WT_FROM_FINALIZE and need_grad and M >= 32768 and W.stride(1) == 1
                            and W.dtype == torch.float32 and not from_gather
                            and not act_from_prev and not (li == nl - 1 and pooled_raw is not None)):
                        # (layers whose backward multiplies by W^T on the Y = A Wt^T kernels: not the
                        # first layer of a grouped stack, not the one-pass and the pooled-algebra layers)
                        Wt = torch.empty((W.shape[1], Cout), device=dev)
                        rec["Wt"] = Wt
                        _call("s2c_bn_finalize_partials_wt", gpart, nbg, M, Cout, gpart.data_ptr(),
                              float(bn.eps), float(mom), _ptr(gamma), _ptr(beta),
                              _ptr(bn.running_mean), _ptr(bn.running_var),
                              scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                              invstd.data_ptr(), _ptr(bn.num_batches_tracked), W.data_ptr(),
                              W.stride(0), W.shape[1], Wt.data_ptr())
                    else:
                        _call("s2c_bn_finalize_partials", gpart, nbg, M, Cout, gpart.data_ptr(),
                              float(bn.eps), float(mom), _ptr(gamma), _ptr(beta),
                              _ptr(bn.running_mean), _ptr(bn.running_var),
                              scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
                              invstd.data_ptr(), _ptr(bn.num_batches_tracked))
                elif train_stats:
                    nb = _stat_blocks(M)
                    if partial is None or partial.numel() < nb * 2 * Cout:
                        partial = torch.empty(nb * 2 * max(Cout, 256), device=dev)
                    mom = bn.momentum if bn.momentum is not None else 0.0
                    _call("s2c_bn_train_stats", Y, M, Cout, Y.data_ptr(),
                          partial.data_ptr(), float(bn.eps), float(mom),
                          _ptr(gamma), _ptr(beta), _ptr(bn.running_mean),
                          _ptr(bn.running_var), scale.data_ptr(), shift.data_ptr(),
                          mean.data_ptr(), invstd.data_ptr(),
                          _ptr(bn.num_batches_tracked), alg_bytes=4 * M * Cout)
                else:
                    _call("s2c_bn_eval_coeffs", Y, Cout, float(bn.eps), _ptr(gamma),
                          _ptr(beta), bn.running_mean.data_ptr(),
                          bn.running_var.data_ptr(), scale.data_ptr(),
                          shift.data_ptr(), mean.data_ptr(), invstd.data_ptr())
                rec.update(Y=Y, scale=scale, shift=shift, mean=mean, invstd=invstd,
                           gamma=gamma, frozen=not train_stats, relu=sp.relu)
                if last and pool_ns > 0 and pooled_raw is not None:
                    J = M // pool_ns
                    out = torch.empty((J, Cout), device=dev)
                    _call("s2c_pool_select", out, J, Cout, pooled_raw[0].data_ptr(),
                          scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
                          alg_bytes=4 * 2 * J * Cout)
                    rec["arg"] = pooled_raw[1]
                    rec["ymax"] = pooled_raw[0]
                    rec["algebra"] = True
                elif last and pool_ns > 0 and pooled_ext is not None:
                    J = M // pool_ns
                    out = torch.empty((J, Cout), device=dev)
                    _call("s2c_pool_select", out, J, Cout, pooled_ext[0].data_ptr(),
                          scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
                          alg_bytes=4 * 2 * J * Cout)
                    rec["arg"] = pooled_ext[1]
                    rec["ymax"] = pooled_ext[0]
                elif last and pool_ns > 0:
                    J = M // pool_ns
                    out = torch.empty((J, Cout), device=dev)
                    arg = torch.empty((J, Cout), dtype=torch.int32, device=dev)
                    ymax = torch.empty((J, Cout), device=dev) if need_grad else None
                    _call("s2c_bn_relu_max", Y, J, pool_ns, Cout, Y.data_ptr(),
                          scale.data_ptr(), shift.data_ptr(), out.data_ptr(),
                          arg.data_ptr(), _ptr(ymax),
                          alg_bytes=4 * (M * Cout + 2 * J * Cout))
                    rec["arg"] = arg
                    rec["ymax"] = ymax
                elif (FUSE_BNRELU_GEMM and not last and _next_takes_prologue(specs, params, pi, li, M, Cout)):
                    deferred = (Y, scale, shift, sp.relu)
                    A = None
                else:
                    A = torch.empty_like(Y)
                    _call("s2c_bn_relu", Y, M, Cout, Y.data_ptr(), scale.data_ptr(),
                          shift.data_ptr(), A.data_ptr(), int(sp.relu),
                          alg_bytes=8 * M * Cout)
                    out = A
            else:
                if sp.relu:
                    rec.update(Y=Y, relu=True)
                    A = Y if pre_activated else torch.relu(Y)
                else:
                    A = Y
                out = A
                if last and pool_ns > 0:
                    raise NotImplementedError("pooling needs a BN+ReLU last layer")
            saved.append(rec)
        if need_grad:
            ctx.saved = saved
            ctx.specs = specs
            ctx.pool_ns = pool_ns
            ctx.gather = gather
            ctx.x_needs_grad = (gather.needs_grad if gather is not None
                                else X.requires_grad)
        return out

    @staticmethod
    def backward(ctx, dOut):
        saved, specs, pool_ns = ctx.saved, ctx.specs, ctx.pool_ns
        dev = dOut.device
        grads = []          # per layer, reversed
        pending = []        # split-K partials of the weight gradients, summed in one launch
        post = []           # closures that finish a gradient once the partial sums exist
        bias_jobs = []      # (dY, layer slot): bias gradients, summed in one launch
        dA = dOut.contiguous()
        partial = None
        nl = len(specs)
        gather = ctx.gather
        prestats = None      # (partial, rows) of THIS layer's BN-backward sums, left by the GEMM
                             # that produced dA (s2c_bn_bwd_gemm_next_stats)
        for li in range(nl - 1, -1, -1):
            rec, sp = saved[li], specs[li]
            W, A_in = rec["W"], rec["A_in"]
            lazy_dw = False
            fused_dA = None
            dW_fused = None
            pre, prestats = prestats, None
            point_grads = None
            if rec.get("act_from_prev") in ("stream", "algebra"):
                pass    # the kernels that read it recompute it (or it is materialised where they do not)
            elif rec.get("act_from_prev"):
                # this layer's input was not kept: it is the previous layer's activation, which the
                # one-pass kernel recomputes (any other branch below gets it materialised here)
                if not (FUSE_BWD_DX_DW and FUSE_BWD_GEMM and BWD_STATS_IN_GEMM and _gemm_split_on()
                        and dA.dtype == torch.float32 and dA.stride(1) == 1
                        and dA.stride(0) == W.shape[0]):
                    A_in = _activation_of(saved[li - 1])
            elif A_in is None:
                if POINT_SPACE_BWD and SCATTER_DW and W.shape[0] % 4 == 0:
                    lazy_dw = True      # weight and input gradients from point-indexed sums
                elif gather.needs_grad or not SCATTER_DW:
                    A_in = gather.materialise()
                else:
                    lazy_dw = True      # dW from point-indexed sums (GatherSpec.weight_grad)
            Cout = W.shape[0]
            M = (A_in.shape[0] if A_in is not None
                 else saved[li - 1]["Y"].shape[0] if rec.get("act_from_prev") else gather.rows)
            dgamma = dbeta = None
            if rec.get("algebra"):
                # pooled last layer without Y3 / dY3 (pooled_layer_backward)
                need_dA = li > 0 or ctx.x_needs_grad
                prev = saved[li - 1] if li > 0 else None
                Kin = W.shape[1]
                nbn = None
                if (BWD_STATS_IN_GEMM and need_dA and prev is not None
                        and specs[li - 1].bn is not None and prev.get("Y") is not None
                        and prev["Y"].shape == (M, Kin) and prev["Y"].is_contiguous()):
                    nbn = prev
                act = None
                if A_in is None and rec.get("act_from_prev") == "algebra":
                    if POOL_ALGEBRA_ACT and DW_STREAM:
                        A_in, act = prev["Y"], (prev["scale"], prev["shift"], prev["relu"])
                    else:
                        A_in = _activation_of(prev)
                dA, dW, dgamma, dbeta = pooled_layer_backward(
                    dA, rec["arg"], rec["ymax"], rec["scale"], rec["shift"], rec["mean"],
                    rec["invstd"], rec["gamma"], rec["frozen"], A_in, W, pool_ns, need_dA=need_dA,
                    next_bn=nbn, act=act)
                if nbn is not None:
                    prestats = nbn.pop("prestats", None)
                g = [dW]
                if sp.bn is not None:
                    g += [dgamma, dbeta]
                grads.append(g)
                continue
            if sp.bn is not None:
                Y = rec["Y"]
                nb = _stat_blocks(M)
                if partial is None or partial.numel() < nb * 2 * Cout:
                    partial = torch.empty(nb * 2 * max(Cout, 256), device=dev)
                coef = torch.empty(3 * Cout, device=dev)
                has_affine = rec["gamma"] is not None
                dgamma = torch.empty(Cout, device=dev) if has_affine else None
                dbeta = torch.empty(Cout, device=dev) if has_affine else None
                lazy_bn = None
                dY = None if (lazy_dw and FUSE_DY_SCATTER and not (li == nl - 1 and pool_ns > 0)
                              and not rec["has_bias"] and dA.stride(1) == 1
                              and dA.stride(0) == Cout) else torch.empty_like(Y)
                def statistics():
                    """coef / dgamma / dbeta of this layer: from the sums the producing GEMM left,
                    or by the statistics pass over (dA, Y)"""
                    if pre is not None:
                        _call("s2c_bn_bwd_finalize_partials", Y, pre[1], M, Cout, pre[0].data_ptr(),
                              int(rec["frozen"]), _ptr(rec["gamma"]), rec["invstd"].data_ptr(),
                              coef.data_ptr(), _ptr(dgamma), _ptr(dbeta))
                        return
                    _call("s2c_bn_relu_bwd_stats", Y, M, Cout, dA.data_ptr(), Y.data_ptr(),
                          rec["scale"].data_ptr(), rec["shift"].data_ptr(),
                          rec["mean"].data_ptr(), rec["invstd"].data_ptr(),
                          _ptr(rec["gamma"]), int(rec["relu"]), int(rec["frozen"]),
                          partial.data_ptr(), coef.data_ptr(), _ptr(dgamma),
                          _ptr(dbeta), alg_bytes=4 * 2 * M * Cout)

                if dY is None:
                    # statistics only; dY itself is formed inside GatherSpec.weight_grad
                    statistics()
                    lazy_bn = (dA, Y, rec["scale"], rec["shift"], rec["mean"], rec["invstd"],
                               coef, int(rec["relu"]))
                elif li == nl - 1 and pool_ns > 0:
                    J = M // pool_ns
                    _call("s2c_bn_relu_max_bwd", Y, J, pool_ns, Cout, dA.data_ptr(),
                          rec["arg"].data_ptr(), rec["ymax"].data_ptr(), Y.data_ptr(),
                          rec["scale"].data_ptr(),
                          rec["shift"].data_ptr(), rec["mean"].data_ptr(),
                          rec["invstd"].data_ptr(), _ptr(rec["gamma"]),
                          int(rec["frozen"]), partial.data_ptr(), coef.data_ptr(),
                          _ptr(dgamma), _ptr(dbeta), dY.data_ptr(),
                          alg_bytes=4 * (2 * M * Cout + 2 * J * Cout))
                elif (FUSE_BWD_GEMM and (li > 0 or (ctx.x_needs_grad and not lazy_dw))
                      and W.dtype == torch.float32
                      and dA.dtype == torch.float32 and dA.stride(1) == 1
                      and dA.stride(0) == Cout and _fused_bwd_pays(M, Cout, W.shape[1])
                      and _gemm_split_on()):
                    # statistics, then dY and the input gradient dX = dY W in ONE pass
                    statistics()
                    Cin = W.shape[1]
                    fused_dA = torch.empty((M, Cin), device=dev)
                    prev = saved[li - 1] if li > 0 else None
                    takes_next = (BWD_STATS_IN_GEMM and prev is not None
                                  and specs[li - 1].bn is not None
                                  and not prev.get("algebra") and prev.get("Y") is not None
                                  and prev["Y"].shape == (M, Cin) and prev["Y"].is_contiguous()
                                  and Cin % 4 == 0)
                    fparts = 0
                    if (FUSE_BWD_DX_DW and takes_next and Cout == 64 and Cin == 64 and not lazy_dw
                            and not rec["has_bias"] and Y.is_contiguous() and W.stride(1) == 1
                            and M >= DW_STREAM_MIN_ROWS and pending is not None):
                        fparts = _bwd_dx_dw_parts(M)
                    if fparts > 0:
                        # ... and the weight gradient too: this layer's input is the previous layer's
                        # activation, recomputed from its pre-activation -- dY is never written
                        npart = torch.empty(fparts * 2 * Cin, device=dev)
                        wpart = torch.empty((fparts, Cout, Cin), device=dev)
                        dW_fused = torch.empty((Cout, Cin), device=dev)
                        dY = None
                        _call("s2c_bn_bwd_dx_dw64", Y, M, dA.data_ptr(), Y.data_ptr(),
                              rec["scale"].data_ptr(), rec["shift"].data_ptr(),
                              rec["mean"].data_ptr(), rec["invstd"].data_ptr(), coef.data_ptr(),
                              int(rec["relu"]), W.data_ptr(), W.stride(0), fused_dA.data_ptr(),
                              prev["Y"].data_ptr(), prev["scale"].data_ptr(),
                              prev["shift"].data_ptr(), prev["mean"].data_ptr(),
                              prev["invstd"].data_ptr(), int(prev["relu"]), wpart.data_ptr(),
                              npart.data_ptr(), alg_bytes=4 * M * (3 * Cout + Cin),
                              alg_flops=4 * M * Cout * Cin)
                        pending.append((wpart, dW_fused))
                        prestats = (npart, fparts)
                    elif takes_next:
                        # ... and the column sums of the PREVIOUS layer's BatchNorm backward out
                        # of the same GEMM's epilogue (its upstream gradient is this output)
                        Wt = rec.get("Wt")
                        if Wt is None:
                            Wt = W.t().contiguous()
                        nbg = _gemm_blocks(M, Cin)
                        npart = torch.empty(nbg * 2 * Cin, device=dev)
                        _call("s2c_bn_bwd_gemm_next_stats", Y, M, Cout, Cin, dA.data_ptr(),
                              Y.data_ptr(), rec["scale"].data_ptr(), rec["shift"].data_ptr(),
                              rec["mean"].data_ptr(), rec["invstd"].data_ptr(), coef.data_ptr(),
                              int(rec["relu"]), Wt.data_ptr(), Wt.stride(0), dY.data_ptr(),
                              fused_dA.data_ptr(), Cin, prev["Y"].data_ptr(),
                              prev["scale"].data_ptr(), prev["shift"].data_ptr(),
                              prev["mean"].data_ptr(), prev["invstd"].data_ptr(),
                              int(prev["relu"]), npart.data_ptr(),
                              alg_bytes=4 * M * (3 * Cout + 2 * Cin),
                              alg_flops=2 * M * Cout * Cin)
                        prestats = (npart, nbg)
                    else:
                        Wt = rec.get("Wt")
                        if Wt is None:
                            Wt = W.t().contiguous()
                        _call("s2c_bn_bwd_gemm", Y, M, Cout, Cin, dA.data_ptr(), Y.data_ptr(),
                              rec["scale"].data_ptr(), rec["shift"].data_ptr(),
                              rec["mean"].data_ptr(), rec["invstd"].data_ptr(), coef.data_ptr(),
                              int(rec["relu"]), Wt.data_ptr(), Wt.stride(0), dY.data_ptr(),
                              fused_dA.data_ptr(), Cin,
                              alg_bytes=4 * M * (3 * Cout + Cin), alg_flops=2 * M * Cout * Cin)
                elif pre is not None and Cout % 4 == 0:
                    statistics()
                    _call("s2c_bn_relu_bwd_apply", Y, M, Cout, dA.data_ptr(), Y.data_ptr(),
                          rec["scale"].data_ptr(), rec["shift"].data_ptr(),
                          rec["mean"].data_ptr(), rec["invstd"].data_ptr(), coef.data_ptr(),
                          int(rec["relu"]), dY.data_ptr(), alg_bytes=4 * 3 * M * Cout)
                else:
                    _call("s2c_bn_relu_bwd", Y, M, Cout, dA.data_ptr(), Y.data_ptr(),
                          rec["scale"].data_ptr(), rec["shift"].data_ptr(),
                          rec["mean"].data_ptr(), rec["invstd"].data_ptr(),
                          _ptr(rec["gamma"]), int(rec["relu"]), int(rec["frozen"]),
                          partial.data_ptr(), coef.data_ptr(), _ptr(dgamma),
                          _ptr(dbeta), dY.data_ptr(), alg_bytes=4 * 5 * M * Cout)
            elif rec.get("relu"):
                dY = torch.ops.aten.threshold_backward(dA, rec["Y"], 0.0)     # dA * (Y > 0), one launch
            else:
                dY = dA
            if lazy_dw:
                dW = gather.weight_grad(dY, bn_bwd=lazy_bn if sp.bn is not None else None,
                                        pending=pending, post=post)
                if gather.needs_grad:
                    point_grads = gather.input_grads(W)
            elif dW_fused is not None:
                dW = dW_fused
            else:
                dW = None
                if A_in is None and rec.get("act_from_prev"):
                    prev = saved[li - 1]
                    if (rec["act_from_prev"] == "stream" and DW_STREAM and dY.is_cuda
                            and dY.dtype == torch.float32 and dY.stride(1) == 1):
                        dW = _weight_grad_stream(dY, prev["Y"], pending,
                                                 act=(prev["scale"], prev["shift"], prev["relu"]))
                    if dW is None:
                        A_in = _activation_of(prev)
                if dW is None:
                    dW = _weight_grad(dY, A_in, pending)
            dbias = None
            if rec["has_bias"]:
                if BATCH_PARTIAL_SUMS and dY.is_cuda and dY.dtype == torch.float32 \
                        and dY.stride(1) == 1:
                    bias_jobs.append((dY, len(grads)))      # summed after the loop
                else:
                    dbias = dY.sum(0)
            need_dA = li > 0 or (ctx.x_needs_grad and not lazy_dw)
            if not need_dA:
                dA = None
            elif fused_dA is not None:
                dA = fused_dA
            elif (dY.is_cuda and dY.dtype == torch.float32
                  and W.dtype == torch.float32 and dY.stride(1) == 1
                  and _hand_da_pays(M, Cout, W.shape[1])):
                Cin = W.shape[1]
                prev = saved[li - 1] if li > 0 else None
                dA = None
                if (BWD_STATS_IN_GEMM and Cin > 64 and prev is not None
                        and specs[li - 1].bn is not None and not prev.get("algebra")
                        and prev.get("Y") is not None and prev["Y"].shape == (M, Cin)
                        and prev["Y"].is_contiguous() and Cin % 4 == 0 and _gemm_split_on()):
                    # the previous layer's BN-backward column sums out of this GEMM's epilogue
                    Wt = rec.get("Wt")
                    if Wt is None:
                        Wt = W.t().contiguous()
                    nbg = _gemm_blocks(M, Cin)
                    npart = torch.empty(nbg * 2 * Cin, device=dev)
                    dX = torch.empty((M, Cin), device=dev)
                    if _C.TIMER.enabled:
                        _C.TIMER.alg_bytes = 4 * M * (Cout + 2 * Cin)
                        _C.TIMER.alg_flops = 2 * M * Cout * Cin
                        _C.TIMER.label = None
                    rc = _C.call("s2c_rows_gemm_next_stats", M, Cin, Cout, dY.data_ptr(),
                                 dY.stride(0), Wt.data_ptr(), Wt.stride(0), dX.data_ptr(),
                                 prev["Y"].data_ptr(), prev["scale"].data_ptr(),
                                 prev["shift"].data_ptr(), prev["mean"].data_ptr(),
                                 prev["invstd"].data_ptr(), int(prev["relu"]), npart.data_ptr(),
                                 _C.stream_ptr(), allow=(-2,))
                    if rc == 0:
                        dA, prestats = dX, (npart, nbg)
                if dA is None:
                    dA = _input_grad_gemm(dY, W, rec.get("Wt"))
            else:
                dA = small_gemm(dY, W, False)
                if dA is None:
                    dA = torch.mm(dY, W)
            g = [dW]
            if rec["has_bias"]:
                g.append(dbias)
            if sp.bn is not None:
                g += [dgamma, dbeta]
            grads.append(g)
        if pending:
            flush_partial_sums(pending)
        for fin in post:
            fin()
        if bias_jobs:
            for (_, slot), db in zip(bias_jobs, row_sums([x for x, _ in bias_jobs])):
                grads[slot][1] = db
        flat = []
        for g in reversed(grads):
            flat += g
        ctx.saved = None
        d_xyz = d_new = d_feats = None
        if gather is not None:
            if point_grads is not None:
                d_xyz, d_new, d_feats = point_grads
            elif dA is not None:
                d_xyz, d_new, d_feats = gather.scatter(dA)
            dA = None
        return (dA, d_xyz, d_new, d_feats, None, None, None) + tuple(flat)


# inference: BN + ReLU (+ max-pool) in the GEMM epilogue (s2c_rows_gemm_bn_eval)
FUSE_EVAL_EPILOGUE = True


def _eval_fusable(specs, params, pool_ns):
    if not (FUSE_EVAL_EPILOGUE and USE_MFMA_GEMM) or pool_ns not in (0, 16, 32, 64):
        return False
    pi = 0
    for sp in specs:
        W = params[pi]
        pi += 1 + (1 if sp.has_bias else 0) + (2 if sp.bn is not None else 0)
        bn = sp.bn
        if (bn is not None and (bn.training or bn.running_mean is None)) or W.stride(1) != 1:
            return False
    return True


def _plain_stack(specs):
    """every layer = conv (no bias) + frozen BN: what the one-kernel inference stage takes"""
    return all(sp.bn is not None and not sp.has_bias for sp in specs)


_EVAL_CONST = {}


def _eval_affine(sp, bias, gamma, beta, Cout, dev):
    """(gamma, beta, mean, var, eps) for the GEMM's inference epilogue out = relu?(acc * sc + sh),
    sc = gamma / sqrt(var + eps), sh = beta - mean * sc.  A bias in front of a frozen BatchNorm
    moves its mean (BN(y + b) = (y - (mean - b)) sc + beta); a layer without BatchNorm is the
    identity affine (gamma = var = 1, eps = 0: sc = 1 exactly) with beta = bias."""
    bn = sp.bn
    if bn is not None:
        mean = bn.running_mean if bias is None else bn.running_mean - bias
        return gamma, beta, mean, bn.running_var, float(bn.eps)
    key = (dev, Cout)
    if key not in _EVAL_CONST:
        _EVAL_CONST[key] = (torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev))
    ones, zeros = _EVAL_CONST[key]
    return ones, (bias if bias is not None else zeros), zeros, ones, 0.0


class _EvalLayer(ctypes.Structure):
    """s2c_eval_layer (include/s2c_fused.h)."""
    _fields_ = [("W", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
                ("mean", ctypes.c_void_p), ("var", ctypes.c_void_p), ("N", ctypes.c_int),
                ("ldw", ctypes.c_int), ("eps", ctypes.c_float)]


_C.register("s2c_sa_fused_eval", [_I, _I, _I, _I, _I, _L, _L, ctypes.c_float, _I, _P, _P, _P, _P, _P,
                                  _P, _I, _P])
# the whole inference stage (gather -> 3 layers -> max) in ONE kernel where its weights fit LDS
# (csrc/s2c_sa_fused.hip: SA1 with few input channels); = False: per-layer kernels
FUSE_EVAL_STAGE = True


def _eval_stage_fused(gather, M, dev, specs, pool_ns, params):
    """-> pooled (M / pool_ns, N3) tensor, or None when the fused kernel does not take the stage."""
    g = gather
    if not (FUSE_EVAL_STAGE and g is not None and len(specs) == 3 and pool_ns == g.ns
            and _plain_stack(specs) and all(sp.relu for sp in specs) and (g.m * g.ns) % 32 == 0):
        return None
    Ws = [params[3 * l] for l in range(3)]
    if [W.shape[1] for W in Ws] != [3 + g.C, Ws[0].shape[0], Ws[1].shape[0]]:
        return None
    lib = _C.load()
    if not getattr(lib, "_fused_eval_sig", False):
        lib.s2c_sa_fused_eval_supported.argtypes = [_I] * 5
        lib.s2c_sa_fused_eval_supported.restype = _I
        lib._fused_eval_sig = True
    if not lib.s2c_sa_fused_eval_supported(g.ns, g.C, *[W.shape[0] for W in Ws]):
        return None
    layers = (_EvalLayer * 3)()
    for l, sp in enumerate(specs):
        W, gamma, beta = params[3 * l], params[3 * l + 1], params[3 * l + 2]
        bn = sp.bn
        layers[l] = _EvalLayer(W.data_ptr(), _ptr(gamma), _ptr(beta), bn.running_mean.data_ptr(),
                               bn.running_var.data_ptr(), W.shape[0], W.stride(0), float(bn.eps))
    N3 = Ws[2].shape[0]
    out = torch.empty((M // pool_ns, N3), device=dev)
    flops = 2 * M * ((3 + g.C) * Ws[0].shape[0] + Ws[0].shape[0] * Ws[1].shape[0]
                     + Ws[1].shape[0] * N3)
    if _C.TIMER.enabled:
        # fused contract: unique source rows + ball-query rows + pooled output
        _C.TIMER.alg_bytes = 4 * (min(g.B * g.N, M) * (3 + g.C) + M + out.numel())
        _C.TIMER.alg_flops = flops
    with torch.cuda.device(dev):
        rc = _C.call("s2c_sa_fused_eval", g.B, g.N, g.m, g.ns, g.C, g.frs, g.fbs, g.radius,
                     g.normalize, g.xyz.data_ptr(), g.new_xyz.data_ptr(), _ptr(g.feats),
                     g.idx.data_ptr(), ctypes.addressof(layers), out.data_ptr(), N3,
                     _C.stream_ptr(), allow=(-2,))
    return out if rc == 0 else None


# inference: the first layer of a gather stack in point space too (= False: the
# gather-fused GEMM of rounds 1-4)
EVAL_POINT_SPACE = True
_C.register("s2c_sa_gather_add_eval", [_I, _I, _I, _I, _I, ctypes.c_float, _I, _P, _P, _P, _P, _P, _I,
                                       _P, _P, _P, _P, ctypes.c_float, _I, _P, _P])


def _eval_stack(X, gather, M, dev, specs, pool_ns, params):
    """Frozen-BN layer stack: one launch per layer, no pre-activation tensor, the last
    layer leaves max-pooled (or the whole stage in one launch: _eval_stage_fused)."""
    fused_out = _eval_stage_fused(gather, M, dev, specs, pool_ns, params)
    if fused_out is not None:
        return fused_out
    A, pi, nl = X, 0, len(specs)
    for li, sp in enumerate(specs):
        W = params[pi]; pi += 1
        bias = gamma = beta = None
        if sp.has_bias:
            bias = params[pi]; pi += 1
        if sp.bn is not None:
            gamma, beta = params[pi], params[pi + 1]; pi += 2
        Cout = W.shape[0]
        gamma, beta, mean, var, eps = _eval_affine(sp, bias, gamma, beta, Cout, dev)
        pn = pool_ns if li == nl - 1 else 0
        out = torch.empty((M // pn if pn else M, Cout), device=dev)
        if (gather is not None and li == 0 and POINT_SPACE and EVAL_POINT_SPACE and pn == 0
                and Cout % 4 == 0 and Cout <= 256 and W.stride(1) == 1
                and gather.feats2d() is not False):
            # the first layer in point space (as in training, section 4.13): the feature product once
            # per POINT, then P[idx] + W_x rel with the frozen BatchNorm + ReLU in the same pass -- no
            # gather-multiply over the B m ns gathered rows
            g = gather
            P = None
            if g.C > 0:
                f2, Wf = g.feats2d(), W[:, 3:]
                P = torch.empty((g.B * g.N, Cout), device=dev)
                _point_gemm(P, f2, Wf, g.B * g.N, Cout, g.C, 4 * min(g.B * g.N, M) * g.C,
                            2 * g.B * g.N * g.C * Cout)
            _call("s2c_sa_gather_add_eval", out, g.B, g.N, g.m, g.ns, Cout, g.radius, g.normalize,
                  g.xyz.data_ptr(), g.new_xyz.data_ptr(), _ptr(P), g.idx.data_ptr(),
                  W.data_ptr(), W.stride(0), _ptr(gamma), _ptr(beta), mean.data_ptr(),
                  var.data_ptr(), eps, int(sp.relu), out.data_ptr(),
                  alg_bytes=4 * (min(g.B * g.N, M) * 3 + M + M * Cout), label="s2c_sa_gather_add")
        elif gather is not None and li == 0:
            g = gather
            _call("s2c_sa_gather_gemm_bn_eval", out, g.B, g.N, g.m, g.ns, g.C, g.frs, g.fbs,
                  g.radius, g.normalize, g.xyz.data_ptr(), g.new_xyz.data_ptr(),
                  _ptr(g.feats), g.idx.data_ptr(), Cout, W.data_ptr(), W.stride(0),
                  _ptr(gamma), _ptr(beta), mean.data_ptr(), var.data_ptr(), eps, int(sp.relu), pn,
                  out.data_ptr(), Cout,
                  alg_bytes=4 * (min(g.B * g.N, M) * (3 + g.C) + M + out.numel()),
                  alg_flops=2 * M * (3 + g.C) * Cout)
        else:
            if A.stride(1) != 1:
                A = A.contiguous()
            K_in = A.shape[1]
            _call("s2c_rows_gemm_bn_eval", out, M, Cout, K_in, A.data_ptr(), A.stride(0),
                  W.data_ptr(), W.stride(0), _ptr(gamma), _ptr(beta), mean.data_ptr(),
                  var.data_ptr(), eps, int(sp.relu), pn, out.data_ptr(), Cout,
                  alg_bytes=4 * (M * K_in + out.numel()), alg_flops=2 * M * K_in * Cout)
        A = out
    return A


# dW = dY^T A by the hand-written kernel (csrc/s2c_dw.hip) instead of a split-K batched
# library GEMM + a partial-sum kernel.  OFF: measured slower (tools/bench_dw.py) --
# its register-fed MFMA loop only matches the library (135 vs 123 us at 1M x 64 x 64) and the
# in-kernel "last workgroup adds up" reduction costs 150-250 us in agent-scope fences.
USE_DW_KERNEL = False
DW_KERNEL_MIN_ROWS = 2048
_dw_counters = {}


def _dw_sizes(M, Cout, Cin):
    lib = _C.load()
    if not getattr(lib, "_dw_sized", False):
        for name in ("s2c_weight_grad_workspace_bytes", "s2c_weight_grad_counter_bytes"):
            fn = getattr(lib, name)
            fn.restype = ctypes.c_longlong
            fn.argtypes = [_L, _I, _I]
        lib._dw_sized = True
    return (lib.s2c_weight_grad_workspace_bytes(M, Cout, Cin),
            lib.s2c_weight_grad_counter_bytes(M, Cout, Cin))


def weight_grad_kernel(dY, A):
    """dW (Cout,Cin) = dY^T A through s2c_weight_grad (fp32 rows with unit column stride)."""
    M, Cout = dY.shape
    Cin = A.shape[1]
    dev = dY.device
    wbytes, cbytes = _dw_sizes(M, Cout, Cin)
    cnt = _dw_counters.get(dev)
    if cnt is None or cnt.numel() * 4 < cbytes:
        # zero once: the kernel leaves its counters zeroed (launches are stream-ordered)
        cnt = torch.zeros(max(cbytes // 4, 4096), dtype=torch.int32, device=dev)
        _dw_counters[dev] = cnt
    work = torch.empty(max(wbytes // 4, 1), dtype=torch.float32, device=dev)
    dW = torch.empty((Cout, Cin), dtype=torch.float32, device=dev)
    _call("s2c_weight_grad", dW, M, Cout, Cin, dY.data_ptr(), dY.stride(0), A.data_ptr(),
          A.stride(0), dW.data_ptr(), Cin, work.data_ptr(), cnt.data_ptr(),
          alg_bytes=4 * M * (Cout + Cin), alg_flops=2 * M * Cout * Cin)
    return dW


DW_MAX_JOBS = 16          # S2C_DW_MAX_JOBS (include/s2c_fused.h)


class _DwJob(ctypes.Structure):
    """s2c_dw_job (include/s2c_fused.h)."""
    _fields_ = [("dY", ctypes.c_void_p), ("A", ctypes.c_void_p), ("dW", ctypes.c_void_p),
                ("part", ctypes.c_void_p), ("M", ctypes.c_longlong), ("ldy", ctypes.c_longlong),
                ("lda", ctypes.c_longlong), ("Cout", ctypes.c_int), ("Cin", ctypes.c_int),
                ("lddw", ctypes.c_int), ("pad_", ctypes.c_int)]


class _DwJobs(ctypes.Structure):
    """s2c_dw_jobs (include/s2c_fused.h)."""
    _fields_ = [("n_jobs", ctypes.c_int), ("pad_", ctypes.c_int), ("job", _DwJob * DW_MAX_JOBS)]


_C.register("s2c_weight_grad_multi", [_P, _P])
# the slab weight gradients of a layer stack (17 us launches that fill a fraction of the chip each) in
# ONE multi-job launch at the stack's flush_partial_sums (= False: one launch per layer)
DW_MULTI = True


class _DeferredDw(object):
    """An entry of a `pending` list: a slab weight gradient not launched yet (flush_partial_sums
    launches all of a stack's together, then adds up their slabs with the other partials).  dY and A
    are kept alive -- and must not be written -- until then."""
    __slots__ = ("dY", "A", "dW", "part")

    def __init__(self, dY, A, dW, part):
        self.dY, self.A, self.dW, self.part = dY, A, dW, part


def _weight_grad_partials(dY, A, pending):
    """dW = dY^T A: the register-fed MFMA kernel of csrc/s2c_dw.hip writes one partial tile per
    row slab; the caller's single multi_colsum launch adds them (kernel boundary instead of
    an in-kernel hand-off)."""
    M, Cout = dY.shape
    Cin = A.shape[1]
    dev = dY.device
    lib = _C.load()
    if not getattr(lib, "_dw_slabs_sized", False):
        lib.s2c_weight_grad_slabs.restype = _I
        lib.s2c_weight_grad_slabs.argtypes = [_L, _I, _I]
        lib._dw_slabs_sized = True
    nslab = lib.s2c_weight_grad_slabs(M, Cout, Cin)
    dW = torch.empty((Cout, Cin), dtype=torch.float32, device=dev)
    part = torch.empty((nslab, Cout, Cin), dtype=torch.float32, device=dev) if nslab > 1 else None
    if DW_MULTI:
        pending.append(_DeferredDw(dY, A, dW, part))
        return dW
    _call("s2c_weight_grad", dW, M, Cout, Cin, dY.data_ptr(), dY.stride(0), A.data_ptr(),
          A.stride(0), dW.data_ptr(), Cin, _ptr(part), None,
          alg_bytes=4 * M * (Cout + Cin), alg_flops=2 * M * Cout * Cin)
    if part is not None:
        pending.append((part, dW))
    return dW


def _launch_deferred_dw(jobs):
    for i in range(0, len(jobs), DW_MAX_JOBS):
        chunk = jobs[i:i + DW_MAX_JOBS]
        a = _DwJobs()
        a.n_jobs = len(chunk)
        nbytes = flops = 0
        for j, d in enumerate(chunk):
            M, Cout = d.dY.shape
            Cin = d.A.shape[1]
            a.job[j] = _DwJob(d.dY.data_ptr(), d.A.data_ptr(), d.dW.data_ptr(), _ptr(d.part), M,
                              d.dY.stride(0), d.A.stride(0), Cout, Cin, Cin, 0)
            nbytes += 4 * M * (Cout + Cin)
            flops += 2 * M * Cout * Cin
        _call("s2c_weight_grad_multi", chunk[0].dW, ctypes.byref(a), alg_bytes=nbytes,
              alg_flops=flops)


COLSUM_MAX_JOBS = 32      # S2C_COLSUM_MAX_JOBS (include/s2c_fused.h)


class _ColsumArgs(ctypes.Structure):
    """s2c_colsum_args (include/s2c_fused.h)."""
    _fields_ = [("n_jobs", ctypes.c_int), ("S", ctypes.c_int * COLSUM_MAX_JOBS),
                ("n", ctypes.c_longlong * COLSUM_MAX_JOBS),
                ("part", ctypes.c_void_p * COLSUM_MAX_JOBS),
                ("out", ctypes.c_void_p * COLSUM_MAX_JOBS),
                ("sub", ctypes.c_void_p * COLSUM_MAX_JOBS),
                ("sub_S", ctypes.c_int * COLSUM_MAX_JOBS), ("ncol", ctypes.c_int * COLSUM_MAX_JOBS),
                ("sub_cols", ctypes.c_int * COLSUM_MAX_JOBS),
                ("sub_div", ctypes.c_float * COLSUM_MAX_JOBS)]


class _ColsumFix(object):
    """An entry of a `pending` list: the first `other.shape[1]` columns of dW become (dW - other) / div
    (div 0: no division) in the partial-sum launch that completes them -- `other` loses its own job."""
    __slots__ = ("dW", "other", "div")

    def __init__(self, dW, other, div):
        self.dW, self.other, self.div = dW, other, float(div)


_C.register("s2c_multi_colsum", [_P, _P])
_C.register("s2c_weight_grad_stream_act", [_L, _I, _I, _P, _L, _P, _L, _P, _P, _I, _P, _P])
BATCH_PARTIAL_SUMS = True


# (Collecting the partial sums of ALL layer stacks of a backward pass into one launch from an
# end-of-backward callback of the autograd engine was built in round 3 and withdrawn: a weight
# gradient handed to AccumulateGrad before it is filled is WRONG whenever the engine clones it
# (another reference alive) or accumulates into an existing .grad (gradient accumulation over
# micro-batches) -- both happen before the callback runs.  ~0.06 ms is not worth that hazard.)


def flush_partial_sums(pending):
    """[(part (S,Cout,Cin), dW (Cout,Cin)) | _DeferredDw ...] -> every dW filled: the deferred slab
    products in one launch per DW_MAX_JOBS, then the sums, COLSUM_MAX_JOBS per launch."""
    jobs = [e for e in pending if isinstance(e, _DeferredDw)]
    if jobs:
        _launch_deferred_dw(jobs)
        pending[:] = [(e.part, e.dW) if isinstance(e, _DeferredDw) else e for e in pending
                      if not (isinstance(e, _DeferredDw) and e.part is None)]
    fixes = [e for e in pending if isinstance(e, _ColsumFix)]
    sums = [e for e in pending if not isinstance(e, _ColsumFix)]
    fix_of = {}
    for f in fixes:
        # the subtrahend's partial table (or, complete already, the tensor itself as one slab) moves into
        # the minuend's job; a minuend written directly by its product gets a one-slab job of its own
        k = next((i for i, (_, o) in enumerate(sums) if o is f.other), None)
        sub = sums.pop(k)[0] if k is not None else f.other.view(1, *f.other.shape)
        if not any(o is f.dW for _, o in sums):
            sums.append((f.dW.view(1, *f.dW.shape), f.dW))
        fix_of[id(f.dW)] = (sub, f)
    for i in range(0, len(sums), COLSUM_MAX_JOBS):
        chunk = sums[i:i + COLSUM_MAX_JOBS]
        a = _ColsumArgs()
        a.n_jobs = len(chunk)
        for j, (part, dW) in enumerate(chunk):
            a.S[j], a.n[j] = part.shape[0], dW.numel()
            a.part[j], a.out[j] = part.data_ptr(), dW.data_ptr()
            fx = fix_of.get(id(dW))
            if fx is not None:
                sub, f = fx
                assert dW.is_contiguous() and sub.is_contiguous() and sub.shape[1] == dW.shape[0]
                a.sub[j], a.sub_S[j] = sub.data_ptr(), sub.shape[0]
                a.ncol[j], a.sub_cols[j], a.sub_div[j] = dW.shape[1], sub.shape[2], f.div
        with torch.cuda.device(chunk[0][1].device):
            _C.call("s2c_multi_colsum", ctypes.byref(a), _C.stream_ptr())
    del pending[:]


class _RowsumArgs(ctypes.Structure):
    """s2c_rowsum_args (include/s2c_fused.h)."""
    _fields_ = [("n_jobs", ctypes.c_int), ("chunk_rows", ctypes.c_int),
                ("C", ctypes.c_int * 16),
                ("M", ctypes.c_longlong * 16), ("ld", ctypes.c_longlong * 16),
                ("X", ctypes.c_void_p * 16), ("out", ctypes.c_void_p * 16)]


_C.register("s2c_multi_rowsum", [_P, _P])


ROWSUM_CHUNK = 1024


def row_sums(mats):
    """[X (M,C) float32, unit column stride, ...] -> [X.sum(0), ...]: one launch per 16
    matrices; tall matrices go through slabs of ROWSUM_CHUNK rows + one partial-sum launch."""
    dev = mats[0].device
    tall = max(x.shape[0] for x in mats) > 2 * ROWSUM_CHUNK
    outs = [torch.empty(x.shape[1], dtype=torch.float32, device=dev) for x in mats]
    pending = []
    for i in range(0, len(mats), 16):
        a = _RowsumArgs()
        a.n_jobs = len(mats[i:i + 16])
        a.chunk_rows = ROWSUM_CHUNK if tall else 0
        for j, (x, o) in enumerate(zip(mats[i:i + 16], outs[i:i + 16])):
            assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
            a.M[j], a.C[j], a.ld[j] = x.shape[0], x.shape[1], x.stride(0)
            a.X[j] = x.data_ptr()
            if tall:
                nslab = (x.shape[0] + ROWSUM_CHUNK - 1) // ROWSUM_CHUNK
                part = torch.empty((nslab, x.shape[1]), dtype=torch.float32, device=dev)
                pending.append((part, o))
                a.out[j] = part.data_ptr()
            else:
                a.out[j] = o.data_ptr()
        with torch.cuda.device(dev):
            _C.call("s2c_multi_rowsum", ctypes.byref(a), _C.stream_ptr())
    if pending:
        flush_partial_sums(pending)
    return outs


_C.register("s2c_weight_grad_stream", [_L, _I, _I, _P, _L, _P, _L, _P, _P])
# tall weight gradients on the streaming kernel (csrc/s2c_dwstream.hip): LDS-DMA ring, column reads
# of the row-major chunks as the transposed MFMA operand, one partial tile per workgroup
DW_STREAM = True
DW_STREAM_MIN_ROWS = 32768


def _dw_stream_parts(M, C, N, dY, A, ldy=None, lda=None):
    lib = _C.load()
    if not getattr(lib, "_dws_sized", False):
        lib.s2c_weight_grad_stream_parts.restype = _I
        lib.s2c_weight_grad_stream_parts.argtypes = [_L, _I, _I, _P, _L, _P, _L]
        lib._dws_sized = True
    return lib.s2c_weight_grad_stream_parts(M, C, N, dY.data_ptr(),
                                            dY.stride(0) if ldy is None else ldy, A.data_ptr(),
                                            A.stride(0) if lda is None else lda)


def _weight_grad_stream(dY, A, pending, act=None):
    """dW (C, N) = dY^T A as per-workgroup partials of s2c_weight_grad_stream (summed by the
    caller's multi_colsum launch).  A: (M, N) rows with unit column stride, any row stride (a
    column block of a wider tensor is read in place).  act = (scale, shift, relu): the operand is
    relu?(A scale + shift), formed on the way (s2c_weight_grad_stream_act).  None: shape not taken."""
    M, C = dY.shape
    N = A.shape[1]
    parts = _dw_stream_parts(M, C, N, dY, A)
    if parts <= 0 or (act is not None and (N <= 16 and C == 64)):
        return None
    dev = dY.device
    part = torch.empty((parts, C, N), dtype=torch.float32, device=dev)
    dW = torch.empty((C, N), dtype=torch.float32, device=dev)
    same = A.data_ptr() == dY.data_ptr() and A.stride(0) == dY.stride(0) and C == N
    if act is not None:
        rc = _C.call("s2c_weight_grad_stream_act", M, C, N, dY.data_ptr(), dY.stride(0), A.data_ptr(),
                     A.stride(0), act[0].data_ptr(), act[1].data_ptr(), int(act[2]), part.data_ptr(),
                     _C.stream_ptr(), allow=(-2,))
        if rc != 0:
            return None
    else:
        _call("s2c_weight_grad_stream", dW, M, C, N, dY.data_ptr(), dY.stride(0), A.data_ptr(),
              A.stride(0), part.data_ptr(),
              alg_bytes=4 * M * (C if same else C + N), alg_flops=2 * M * C * N)
    pending.append((part, dW))
    return dW


# A layer's input activation kept by the forward ONLY for its weight gradient (the side output of the
# BatchNorm + ReLU prologue GEMMs) is not written where that gradient runs on the streaming kernel: the
# kernel recomputes it from the previous layer's pre-activation (s2c_weight_grad_stream_act)
DW_STREAM_ACT = True
# ... and the pooled-layer algebra's three readers of its input (the pool-backward GEMM, the Gram matrix,
# the SP sums) form it from the previous layer's pre-activation too
POOL_ALGEBRA_ACT = True


def _dw_will_stream(M, C, N, ref):
    """Does _weight_grad send (M, C, N) to the streaming kernel?  (its dispatch, in its order; `ref`: a
    tensor with the alignment the operands will have)"""
    return bool(DW_STREAM and DW_STREAM_ACT and BATCH_PARTIAL_SUMS and not USE_DW_KERNEL
                and not _hand_dw_pays(M, C, N) and M >= DW_STREAM_MIN_ROWS and not (N <= 16 and C == 64)
                and _dw_stream_parts(M, C, N, ref, ref, ldy=C, lda=N) > 0)


def _weight_grad(dY, A, pending=None):
    """dW (Cout,Cin) = dY^T (Cout,M) @ A (M,Cin) with M up to ~1e6 and a tiny
    output: a plain GEMM call gives the library ONE output tile and a million-deep
    K loop (a single workgroup).  Hand-written kernel (USE_DW_KERNEL); library fallback
    for exotic layouts: split the row dimension into S independent slabs (strided-batched
    GEMM fills the chip) and reduce the S partials."""
    M = dY.shape[0]
    if (USE_DW_KERNEL and dY.is_cuda and M >= DW_KERNEL_MIN_ROWS and dY.dtype == torch.float32
            and A.dtype == torch.float32 and dY.stride(1) == 1 and A.stride(1) == 1):
        return weight_grad_kernel(dY, A)
    if (dY.is_cuda and dY.dtype == torch.float32 and A.dtype == torch.float32
            and dY.stride(1) == 1 and A.stride(1) == 1 and pending is not None
            and BATCH_PARTIAL_SUMS and _hand_dw_pays(M, dY.shape[1], A.shape[1])):
        return _weight_grad_partials(dY, A, pending)
    if (DW_STREAM and M >= DW_STREAM_MIN_ROWS and dY.is_cuda and dY.dtype == torch.float32
            and A.dtype == torch.float32 and dY.stride(1) == 1 and A.stride(1) == 1
            and pending is not None and BATCH_PARTIAL_SUMS):
        dW = _weight_grad_stream(dY, A, pending)
        if dW is not None:
            return dW
    # slabs of >= 1024 rows (>= 2048 from 256k rows on), at most 256 of them: measured best
    # trade between the batched GEMM and the partial sum (tools/bench_dw_split.py)
    S, rows = 1, (2048 if M >= 262144 else 1024)
    while S < 256 and M % (2 * S) == 0 and M // (2 * S) >= rows:
        S *= 2
    if S == 1:
        return torch.mm(dY.t(), A)
    part = torch.bmm(dY.view(S, M // S, -1).transpose(1, 2), A.view(S, M // S, -1))
    if pending is not None and BATCH_PARTIAL_SUMS and part.is_cuda \
            and part.dtype == torch.float32:
        # the caller adds up the slabs of all its layers in one launch (flush_partial_sums)
        dW = torch.empty(part.shape[1:], dtype=torch.float32, device=part.device)
        pending.append((part, dW))
        return dW
    return part.sum(0)


class GatherSpec(object):
    """Describes the operand of a set-abstraction stack without building it:
    row (b, j, s) = [ (xyz[b, idx] - new_xyz[b, j]) (/radius) | feats[b, idx, :] ]."""

    def __init__(self, xyz, new_xyz, feats, idx, radius, normalize):
        self.xyz, self.new_xyz = xyz.contiguous(), new_xyz.contiguous()
        if feats is not None and feats.stride(2) != 1:
            feats = feats.contiguous()
        self.feats, self.idx = feats, idx
        self.B, self.N = xyz.shape[0], xyz.shape[1]
        self.m, self.ns = idx.shape[1], idx.shape[2]
        self.C = feats.shape[2] if feats is not None else 0
        self.frs = feats.stride(1) if feats is not None else 0
        self.fbs = feats.stride(0) if feats is not None else 0
        self.radius, self.normalize = float(radius), int(bool(normalize))
        self.rows = self.B * self.m * self.ns
        self.needs_grad = self.need_xyz = self.need_feats = False

    def feats2d(self):
        """The features as (B n, C) rows with one row stride (a view); None without features,
        False when the batch stride does not continue the row stride."""
        f = self.feats
        if f is None or self.C == 0:
            return None
        if f.stride(0) != f.shape[1] * f.stride(1):
            return False
        return f.as_strided((self.B * self.N, self.C), (f.stride(1), 1), f.storage_offset())

    def input_grads(self, W):
        """(d_xyz, d_new_xyz, d_feats) from the point-indexed sums weight_grad() left (Z: per
        point, S: per centre): the products with W run over B n + B m rows, not over the gathered
        rows, and nothing is scattered afterwards."""
        ZS = self._ZS
        nz = self.B * self.N
        d_xyz = d_new = d_feats = None
        if self.need_feats and self.C > 0:
            d_feats = _input_grad_gemm(ZS[:nz], W[:, 3:]).view(self.B, self.N, self.C)
        if self.need_xyz:
            g3 = _input_grad_gemm(ZS, W[:, :3])
            if self.normalize:
                g3 = g3 / self.radius
            d_xyz = g3[:nz].view(self.B, self.N, 3)
            d_new = (-g3[nz:]).view(self.B, self.m, 3)
        self._ZS = None
        return d_xyz, d_new, d_feats

    def materialise(self):
        X = torch.empty((self.rows, 3 + self.C), dtype=torch.float32,
                        device=self.xyz.device)
        _call("s2c_sa_gather_rows", X, self.B, self.N, self.m, self.ns, self.C,
              self.frs, self.fbs, self.radius, self.normalize, self.xyz.data_ptr(),
              self.new_xyz.data_ptr(), _ptr(self.feats), self.idx.data_ptr(),
              X.data_ptr(),
              alg_bytes=4 * (min(self.B * self.N, self.rows) * (3 + self.C)
                             + self.rows + self.rows * (3 + self.C)))
        return X

    def weight_grad(self, dY, bn_bwd=None, pending=None, post=None):
        """dW (Cout, 3+C) = dY^T G without building G (csrc/s2c_sa.hip:
        sa_scatter_sum): the products run over the B*N points.  bn_bwd = (dA, Y, scale,
        shift, mean, invstd, coef, relu): dY is None and formed on the fly as the
        BatchNorm(+ReLU) backward of dA (s2c_sa_scatter_sum_bn_bwd).  pending / post: the
        caller's list of split-K partials (flush_partial_sums) and of closures to run after it --
        the returned dW is complete once both have run."""
        ref = dY if dY is not None else bn_bwd[0]
        dev = ref.device
        Cout = ref.shape[1]
        nz = self.B * self.N
        # Z and S back to back: input_grads() multiplies both by W_x in one product
        ZS = torch.empty((nz + self.B * self.m, Cout), device=dev)
        Z = ZS[:nz].view(self.B, self.N, Cout)
        S = ZS[nz:].view(self.B, self.m, Cout)
        self._ZS = ZS
        if bn_bwd is not None and dY is None:
            dA, Y, scale, shift, mean, invstd, coef, relu = bn_bwd
            _call("s2c_sa_scatter_sum_bn_bwd", dA, self.B, self.N, self.m, self.ns, Cout,
                  dA.data_ptr(), Y.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                  mean.data_ptr(), invstd.data_ptr(), coef.data_ptr(), relu,
                  self.idx.data_ptr(), Z.data_ptr(), S.data_ptr(),
                  alg_bytes=4 * (self.rows * (2 * Cout + 1)
                                 + (self.B * self.N + self.B * self.m) * Cout))
        else:
            dY = dY.contiguous()
            _call("s2c_sa_scatter_sum", dY, self.B, self.N, self.m, self.ns, Cout,
                  dY.data_ptr(), self.idx.data_ptr(), Z.data_ptr(), S.data_ptr(),
                  alg_bytes=4 * (self.rows * (Cout + 1)
                                 + (self.B * self.N + self.B * self.m) * Cout))
        Z2, S2 = ZS[:nz], ZS[nz:]
        f2 = self.feats2d()
        if f2 is False:
            f2 = self.feats.reshape(nz, self.C)
        # the centres' share of the coordinate columns: - S^T new_xyz
        dWc = _weight_grad(S2, self.new_xyz.view(-1, 3), pending)
        cloud = getattr(self.xyz, "_s2c_cloud", None)
        rows = None
        if (DW_STREAM and cloud is not None and f2 is not None and f2 is not False and nz > 65536
                and cloud.is_contiguous() and cloud.shape[-1] == 3 + self.C
                and cloud.numel() == nz * (3 + self.C) and f2.stride(0) == 3 + self.C
                and f2.data_ptr() == cloud.data_ptr() + 12):
            rows = cloud.view(nz, 3 + self.C)
        if rows is not None and _dw_stream_parts(nz, Cout, 3 + self.C, Z2, rows) > 0:
            # the features are the cloud's own columns 3.. and xyz its columns 0..2 (set by the
            # backbone, models/backbone_module.py): ONE streaming product over the rows as they lie
            # in memory instead of a 3-column product and a features product
            dW = _weight_grad(Z2, rows, pending)
            dWp = dWf = None
        elif f2 is not None and nz <= 65536:
            # a small stage: ONE product over the points with [xyz | feats] side by side (the copy
            # is <= 35 MB), instead of a 3-column product of its own; its output IS dW
            dW = _weight_grad(Z2, torch.cat([self.xyz.view(-1, 3), f2], 1), pending)
            dWp = dWf = None
        else:
            dW = torch.empty((Cout, 3 + self.C), device=dev)
            dWp = _weight_grad(Z2, self.xyz.view(-1, 3), pending)
            dWf = _weight_grad(Z2, f2, pending) if f2 is not None else None

        if (dWp is None and pending is not None and post is not None and BATCH_PARTIAL_SUMS
                and dW.is_contiguous() and dWc.is_contiguous()):
            # (dW[:, :3] - dWc) / radius leaves with the partial-sum launch that completes dW
            pending.append(_ColsumFix(dW, dWc, self.radius if self.normalize else 0.0))
            return dW

        def finish():
            x3 = dW[:, :3]                       # in-place ops on the view: no copy back
            if dWp is None:
                x3.sub_(dWc)
            else:
                torch.sub(dWp, dWc, out=x3)
                if dWf is not None:
                    dW[:, 3:].copy_(dWf)
            if self.normalize:
                x3.div_(self.radius)
        if post is not None and pending is not None:
            post.append(finish)
        else:
            finish()
        return dW

    def scatter(self, dX):
        """Row gradients (rows, 3+C) -> d_xyz (B,N,3), d_new_xyz (B,m,3),
        d_feats (B,N,C) (None where not needed)."""
        dev = dX.device
        dX = dX.contiguous()
        d_feats = torch.empty((self.B, self.N, self.C), device=dev) \
            if (self.need_feats and self.C > 0) else None
        d_xyz = torch.empty((self.B, self.N, 3), device=dev) if self.need_xyz else None
        d_new = torch.empty((self.B, self.m, 3), device=dev) if self.need_xyz else None
        if d_feats is not None or d_xyz is not None:
            _call("s2c_sa_scatter_rows", dX, self.B, self.N, self.m, self.ns, self.C,
                  self.radius, self.normalize, dX.data_ptr(), self.idx.data_ptr(),
                  _ptr(d_feats), _ptr(d_xyz), _ptr(d_new),
                  alg_bytes=4 * (self.rows * (3 + self.C + 1) + self.B * self.N * self.C))
        return d_xyz, d_new, d_feats


def _layer_params(conv_w, conv_b, bn):
    W = conv_w.view(conv_w.shape[0], -1)
    p = [W]
    if conv_b is not None:
        p.append(conv_b)
    if bn is not None:
        p += [bn.weight, bn.bias]
    return p


def shared_mlp_specs(mlp):
    """(specs, params) of a pointnet2 SharedMLP (layer{i}.conv / .bn.bn / ReLU)."""
    specs, params = [], []
    for layer in mlp.children():
        conv = layer.conv
        bn = layer.bn.bn if hasattr(layer, "bn") else None
        relu = hasattr(layer, "activation")
        specs.append(LayerSpec(conv.bias is not None, bn, relu))
        params += _layer_params(conv.weight, conv.bias, bn)
    return specs, params


def mlp_supported(specs, params):
    """BN kernels need channel counts that are multiples of 4."""
    if not ENABLED:
        return False
    pi = 0
    for sp in specs:
        W = params[pi]
        pi += 1 + (1 if sp.has_bias else 0) + (2 if sp.bn is not None else 0)
        if sp.bn is not None and W.shape[0] % 4 != 0:
            return False
        if sp.bn is not None and sp.bn.momentum is None:
            return False        # cumulative moving average: torch path
    return True


def _no_backward(*tensors):
    """True when nothing downstream can ask for a gradient (torch.no_grad(), or no
    input / parameter requires one).  ctx.needs_input_grad inside a Function ignores the
    grad mode, so this is decided before Function.apply."""
    if not torch.is_grad_enabled():
        return True
    return not any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors)


def mlp_rows(X, specs, params, pool_ns=0):
    """Apply the layer stack to row-major X (M, Cin) -> (M or M/pool_ns, Cout)."""
    if _no_backward(X, *params) and _eval_fusable(specs, params, pool_ns):
        return _eval_stack(X, None, X.shape[0], X.device, specs, pool_ns, params)
    return _MLPRows.apply(X, None, None, None, None, specs, pool_ns, *params)


# fuse the grouping into the first layer's MFMA GEMM (no (rows, 3+C) tensor)
FUSE_GATHER = True


def sa_group_mlp_pool(xyz, new_xyz, feats_pm, idx, radius, normalize, mlp):
    """One set-abstraction stage on point-major data.

    xyz (B,N,3), new_xyz (B,m,3), feats_pm (B,N,C) or None, idx (B,m,ns) int32,
    mlp: pointnet2 SharedMLP.  Returns pooled features (B, m, Cout) point-major.
    """
    B, m, ns = idx.shape
    specs, params = shared_mlp_specs(mlp)
    first_bn = specs[0].bn
    if (FUSE_GATHER and USE_MFMA_GEMM and not specs[0].has_bias
            and params[0].stride(1) == 1):
        if _no_backward(xyz, new_xyz, feats_pm, *params) and _eval_fusable(specs, params, ns):
            g = GatherSpec(xyz, new_xyz, feats_pm, idx, radius, normalize)
            return _eval_stack(None, g, g.rows, xyz.device, specs, ns, params).view(B, m, -1)
        out = _MLPRows.apply(None, xyz, new_xyz, feats_pm, (idx, radius, normalize),
                             specs, ns, *params)
    else:
        X = _GatherRows.apply(xyz, new_xyz, feats_pm, idx, radius, normalize)
        out = mlp_rows(X, specs, params, pool_ns=ns)
    return out.view(B, m, -1)
