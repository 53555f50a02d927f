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
    "s2c_gather_points": [_INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_gather_points_grad": [_INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_ball_query": [_INT, _INT, _INT, _FLT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_group_points": [_INT, _INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_group_points_grad": [_INT, _INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR],
    "s2c_three_nn": [_INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR, _PTR],
    "s2c_three_interpolate": [_INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR, _PTR],
    "s2c_three_interpolate_grad": [_INT, _INT, _INT, _INT, _PTR, _PTR, _PTR, _PTR, _PTR],
}


class S2CError(RuntimeError):
    pass


def lib_path():
    return _build.LIB_PATH


def load():
    """Load (once) and return the ctypes handle.  Raises if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB_PATH
    if not os.path.exists(path):
        raise S2CError(
            "libs2c_hip.so not found at %s -- run `python -m scan2cap_amd.build` "
            "(there is no CPU fallback for the hot path)" % path)
    lib = ctypes.CDLL(path)
    lib.s2c_abi_version.restype = _INT
    lib.s2c_last_error_string.restype = ctypes.c_char_p
    lib.s2c_fps_resident_limit.restype = _INT
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _INT
    _lib = lib
    return lib


def declared_symbols():
    return ["s2c_abi_version", "s2c_last_error_string",
            "s2c_fps_resident_limit"] + list(_SIGNATURES)


def register(name, argtypes):
    """Declare an additional entry point (fused-path kernels add theirs)."""
    _SIGNATURES[name] = argtypes
    if _lib is not None:
        fn = getattr(_lib, name)
        fn.argtypes = argtypes
        fn.restype = _INT


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise S2CError("%s failed (rc=%d): %s" %
                       (name, rc, lib.s2c_last_error_string().decode()))


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
