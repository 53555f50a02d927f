"""Many small fp32 GEMMs in one launch (csrc/s2c_mgemm.hip): the hoisted, recurrence-free products
of the teacher-forced caption decoder (models/decoder_fused.py).  Jobs with ksplit > 1 add their
k-slices with float atomics: those outputs (the decoder's dH2 with ksplit = 16) differ in the last
bits from run to run; ksplit <= 1 is deterministic.  Operands are described by index
maps, not copied: a transposed weight, a column block of a larger matrix or the (t, r) / (r, t) row
orders of the decoder's tensors are strides."""
import ctypes

import torch

from . import _C

_I, _P = ctypes.c_int, ctypes.c_void_p
MAX_JOBS = 32


class _Axis(ctypes.Structure):
    """include/s2c_fused.h: s2c_mgemm_axis"""
    _fields_ = [("div", _I), ("hi", _I), ("lo", _I)]


class _Job(ctypes.Structure):
    """include/s2c_fused.h: s2c_mgemm_job"""
    _fields_ = [("A", _P), ("B", _P), ("C", _P), ("bias", _P), ("M", _I), ("N", _I), ("K", _I),
                ("am", _Axis), ("ak", _Axis), ("bk", _Axis), ("bn", _Axis), ("cm", _Axis),
                ("accumulate", _I), ("ksplit", _I), ("tile0", _I), ("pad_", _I)]


class _Args(ctypes.Structure):
    """include/s2c_fused.h: s2c_mgemm_args"""
    _fields_ = [("n_jobs", _I), ("pad_", _I), ("job", _Job * MAX_JOBS)]


_C.register("s2c_mgemm", [_P, _P])


def ax(lo, div=0, hi=0):
    """index i -> offset: i * lo, or (two tensor dims walked major-first) (i // div) * hi + (i % div) * lo"""
    return (int(div), int(hi), int(lo))


def _reach(axis, n):
    """largest offset an index map produces over 0 .. n-1"""
    div, hi, lo = axis
    if n <= 0:
        return 0
    if div:
        return ((n - 1) // div) * hi + min(n - 1, div - 1) * lo
    return (n - 1) * lo


class Job(object):
    """C (M x N; element (m, n) at C[ix(m, cm) + n]) = A B (+ bias) (+ C): A(m, k) at
    A[ix(m, am) + ix(k, ak)], B(k, n) at B[ix(k, bk) + ix(n, bn)]; A, B, C: tensors whose
    data_ptr() is the origin."""

    def __init__(self, M, N, K, A, am, ak, B, bk, bn, C, cm, bias=None, accumulate=False, ksplit=0):
        for t in (A, B, C):
            assert t.dtype == torch.float32 and t.is_cuda
        self.dims, self.t = (M, N, K), (A, B, C, bias)
        self.axes = (am, ak, bk, bn, cm)
        self.accumulate, self.ksplit = int(accumulate), int(ksplit)


def mm(A, B, C, bias=None, accumulate=False, ksplit=0):
    """C = A @ B for 2-D VIEWS (any strides; C with unit column stride)."""
    M, K = A.shape
    K2, N = B.shape
    assert K == K2 and C.shape == (M, N) and C.stride(1) == 1
    return Job(M, N, K, A, ax(A.stride(0)), ax(A.stride(1)), B, ax(B.stride(0)), ax(B.stride(1)),
               C, ax(C.stride(0)), bias, accumulate, ksplit)


def launch(jobs):
    """All jobs in one launch (MAX_JOBS per launch).  They must not read each other's outputs."""
    for i in range(0, len(jobs), MAX_JOBS):
        chunk = jobs[i:i + MAX_JOBS]
        a = _Args()
        a.n_jobs = len(chunk)
        flops = 0.0
        for j, jb in enumerate(chunk):
            d = a.job[j]
            d.M, d.N, d.K = jb.dims
            A, B, C, bias = jb.t
            d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
            d.bias = bias.data_ptr() if bias is not None else None
            for name, v in zip(("am", "ak", "bk", "bn", "cm"), jb.axes):
                x = getattr(d, name)
                # 32-bit fields on the device side (include/s2c_fused.h): ctypes would truncate silently
                assert all(0 <= int(q) < 2 ** 31 for q in v), (name, v)
                x.div, x.hi, x.lo = v
            M_, N_, K_ = jb.dims
            am, ak, bk, bn, cm = jb.axes
            for what, reach in (("A", _reach(am, M_) + _reach(ak, K_)),
                                ("B", _reach(bk, K_) + _reach(bn, N_)), ("C", _reach(cm, M_) + N_)):
                assert reach < 2 ** 31, "s2c_mgemm: operand %s reaches element %d (>= 2^31)" % (what, reach)
            d.accumulate, d.ksplit = jb.accumulate, jb.ksplit
            flops += 2.0 * d.M * d.N * d.K
        if _C.TIMER.enabled:
            _C.TIMER.alg_flops = flops
        with torch.cuda.device(chunk[0].t[2].device):
            _C.call("s2c_mgemm", ctypes.byref(a), _C.stream_ptr())


# ---- the decoder's three classifier products on the matrix cores (csrc/s2c_sgemm.hip) -----------------
class _SgMap(ctypes.Structure):
    """include/s2c_fused.h: s2c_sgemm_map"""
    _fields_ = [("div", _I), ("hi", _I), ("lo", _I)]


class _SgArgs(ctypes.Structure):
    """include/s2c_fused.h: s2c_sgemm_args"""
    _fields_ = [("A", _P), ("B", _P), ("bias", _P), ("Y", _P), ("M", ctypes.c_longlong), ("N", _I),
                ("K", _I), ("form", _I), ("ksplit", _I), ("arow", _SgMap), ("brow", _SgMap),
                ("crow", _SgMap), ("pad_", _I)]


_C.register("s2c_small_gemm_ex", [_P, _P])
NN, NT, TN = 0, 1, 2


def mfma(form, M, N, K, A, arow, B, brow, Y, crow, bias=None, ksplit=1):
    """Y = A B (NN) | A W^T (NT) | A^T B (TN) on s2c_small_gemm_ex (bf16x3 MFMA, fp32-accurate); the
    row maps are ax(...) triples; A, B, Y: tensors whose data_ptr() is the origin.  ksplit > 1 adds
    its k-ranges into a ZEROED Y with float atomics (not deterministic, like s2c_mgemm's ksplit).
    Returns False when the layout is not taken (the caller keeps its s2c_mgemm job)."""
    for t in (A, B, Y):
        if not (t.is_cuda and t.dtype == torch.float32):
            return False
    if K < 4 or (form != NT and N % 4) or (form == TN and M % 4):
        return False
    if A.data_ptr() % (16 if form == TN else 4) or B.data_ptr() % (4 if form == NT else 16):
        return False
    if form == TN and (arow[1] % 4 or arow[2] % 4):
        return False
    if form != NT and (brow[1] % 4 or brow[2] % 4):
        return False
    lim = 2 ** 29
    if (_reach(arow, K if form == TN else M) + (M if form == TN else K) >= lim
            or _reach(brow, N if form == NT else K) + (K if form == NT else N) >= lim
            or _reach(crow, M) + N >= lim):
        return False
    a = _SgArgs()
    a.A, a.B, a.Y = A.data_ptr(), B.data_ptr(), Y.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.M, a.N, a.K, a.form, a.ksplit = M, N, K, form, int(ksplit)
    for name, v in (("arow", arow), ("brow", brow), ("crow", crow)):
        m = getattr(a, name)
        m.div, m.hi, m.lo = v
    if _C.TIMER.enabled:
        _C.TIMER.alg_flops = 2.0 * M * N * K
        _C.TIMER.alg_bytes = 4 * (M * K + K * N + M * N)
    with torch.cuda.device(Y.device):
        _C.call("s2c_small_gemm_ex", ctypes.byref(a), _C.stream_ptr())
    return True
