"""ctypes binding of libs2c_hip.so (C ABI declared in include/s2c_ops.h).

There is NO CPU fallback: if the HIP library is missing or a launch fails this
module raises.  Tensors are passed as raw device pointers + sizes; work is
enqueued on torch's current HIP stream (the reference uses
at::cuda::getCurrentCUDAStream(), e.g. ball_query_gpu.cu:49).
"""
import ctypes
import os

from . import build as _build

_lib = None

_INT = ctypes.c_int
_PTR = ctypes.c_void_p
_FLT = ctypes.c_float

# name -> argtypes (all return int)
_SIGNATURES = {
    "s2c_furthest_point_sampling": [_INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_furthest_point_sampling_bucketed": [_INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_furthest_point_sampling_small": [_INT, _INT, _INT, _PTR, _PTR, _INT, _PTR],
    "s2c_furthest_point_sampling_cells": [_INT, _INT, _INT, _PTR, _PTR, _PTR, _INT, _PTR],
    "s2c_furthest_point_sampling_prefix": [_INT, _INT, _INT, _PTR, _PTR, _PTR, _INT, _PTR],
    "s2c_gather_points": [_INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_gather_points_grad": [_INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_ball_query": [_INT, _INT, _INT, _FLT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_ball_query_grid": [_INT, _INT, _INT, _FLT, _INT, _PTR, _PTR, _PTR, _PTR, _PTR],
    "s2c_group_points": [_INT, _INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_group_points_grad": [_INT, _INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_three_nn": [_INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR, _PTR],
    "s2c_three_interpolate": [_INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR, _PTR],
    "s2c_three_interpolate_grad": [_INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR, _PTR],
}


class S2CError(RuntimeError):
    pass


def lib_path():
    return _build.lib_path()


def load():
    """Load (once) and return the ctypes handle.  Raises if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.lib_path()      # S2C_NVCC_CONTRACT=1|2: the nvcc-contraction variant
    if not os.path.exists(path):
        raise S2CError(
            "libs2c_hip.so not found at %s -- run `python -m scan2cap_amd.build` "
            "(there is no CPU fallback for the hot path)" % path)
    lib = ctypes.CDLL(path)
    lib.s2c_abi_version.restype = _INT
    lib.s2c_last_error_string.restype = ctypes.c_char_p
    lib.s2c_fps_resident_limit.restype = _INT
    lib.s2c_fps_workspace_bytes.restype = ctypes.c_longlong
    lib.s2c_fps_workspace_bytes.argtypes = [_INT, _INT]
    lib.s2c_fps_cells_workspace_bytes.restype = ctypes.c_longlong
    lib.s2c_fps_cells_workspace_bytes.argtypes = [_INT, _INT]
    lib.s2c_fps_prefix_workspace_bytes.restype = ctypes.c_longlong
    lib.s2c_fps_prefix_workspace_bytes.argtypes = [_INT, _INT]
    lib.s2c_ball_query_workspace_bytes.restype = ctypes.c_longlong
    lib.s2c_ball_query_workspace_bytes.argtypes = [_INT, _INT]
    lib.s2c_ball_query_grid_max_nsample.restype = _INT
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _INT
    _lib = lib
    return lib


def declared_symbols():
    return ["s2c_abi_version", "s2c_last_error_string",
            "s2c_fps_resident_limit", "s2c_fps_workspace_bytes",
            "s2c_fps_small_limit", "s2c_ball_query_workspace_bytes", "s2c_fps_cells_workspace_bytes",
            "s2c_fps_prefix_workspace_bytes", "s2c_ball_query_grid_max_nsample"] + list(_SIGNATURES)


def register(name, argtypes):
    """Declare an additional entry point (fused-path kernels add theirs)."""
    _SIGNATURES[name] = argtypes
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.argtypes = argtypes
        fn.restype = _INT


class KernelTimer(object):
    """Optional per-entry-point timing with HIP events recorded on the stream the
    kernels are launched on (torch's current stream).  Enabled by bench.py over
    its timed region; zero cost when off."""

    def __init__(self):
        self.enabled = False
        self.records = {}      # name -> list of (start_event, end_event, alg_bytes)
        self.alg_bytes = 0     # set by the op wrapper right before call()
        self.alg_flops = 0     # ditto (GEMM entry points)
        self.label = None      # ditto: file the next call under this name instead of the entry point's

    def start(self):
        self.records = {}
        self.enabled = True

    def stop(self):
        """Synchronise and return {name: {"calls", "total_ms", "alg_bytes",
        "max_ms", "max_call_bytes"}}."""
        import torch
        self.enabled = False
        torch.cuda.synchronize()
        # An event pair around NOTHING measures ~5.6 us on this stack (two timestamp
        # packets); around a kernel the excess over the rocprofv3 duration is about half
        # of that (2-3 us per call, checked against profiles/r01k_*: small_linear 8.2 vs
        # 5.1 us, bn_relu_bwd 66.6 vs 64.4 us).  Subtract it, so that entry points made of
        # hundreds of 5 us launches are not over-weighted against the streaming kernels.
        pairs = []
        for _ in range(64):
            s = torch.cuda.Event(enable_timing=True)
            e = torch.cuda.Event(enable_timing=True)
            s.record()
            e.record()
            pairs.append((s, e))
        torch.cuda.synchronize()
        base = 0.5 * sorted(s.elapsed_time(e) for s, e in pairs)[len(pairs) // 2]
        out = {}
        for name, evs in self.records.items():
            tot, nbytes, nflops, worst = 0.0, 0, 0, (0.0, 0)
            for s, e, ab, af in evs:
                ms = max(s.elapsed_time(e) - base, 0.0)
                tot += ms
                nbytes += ab
                nflops += af
                if ms > worst[0]:
                    worst = (ms, ab)
            out[name] = {"calls": len(evs), "total_ms": tot, "alg_bytes": nbytes,
                         "alg_flops": nflops, "max_ms": worst[0],
                         "max_call_bytes": worst[1]}
        self.records = {}
        return out


TIMER = KernelTimer()


def call(name, *args, allow=()):
    """Call an entry point; a non-zero return code raises S2CError unless it is listed in
    `allow` (e.g. S2C_ENOSUP = -2 where the caller has another kernel to fall back to), in
    which case it is returned."""
    lib = load()
    if TIMER.enabled:
        import torch
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(lib, name)(*args)
        e.record()
        TIMER.records.setdefault(TIMER.label or name, []).append((s, e, TIMER.alg_bytes, TIMER.alg_flops))
        TIMER.alg_bytes = 0
        TIMER.alg_flops = 0
        TIMER.label = None
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0 and rc not in allow:
        raise S2CError("%s failed (rc=%d): %s" %
                       (name, rc, lib.s2c_last_error_string().decode()))
    return rc


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
