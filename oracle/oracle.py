"""ctypes front-end of the CPU parity oracle (oracle/s2c_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from scan2cap_amd/ (the product path).

Every function takes / returns numpy arrays with the reference's `_ext`
layouts (lib/pointnet2/_ext_src/src/bindings.cpp:6-19):

    furthest_point_sampling(xyz (B,N,3) f32, m)             -> (B,m) i32
    gather_points(points (B,C,N) f32, idx (B,m) i32)        -> (B,C,m) f32
    gather_points_grad(grad (B,C,m), idx (B,m), n)          -> (B,C,n)
    ball_query(new_xyz (B,m,3), xyz (B,N,3), radius, ns)    -> (B,m,ns) i32
    group_points(points (B,C,N), idx (B,m,ns))              -> (B,C,m,ns)
    group_points_grad(grad (B,C,m,ns), idx, n)              -> (B,C,n)
    three_nn(unknown (B,n,3), known (B,m,3))                -> dist2 (B,n,3) f32, idx (B,n,3) i32
    three_interpolate(points (B,C,m), idx (B,n,3), w (B,n,3)) -> (B,C,n)
    three_interpolate_grad(grad (B,C,n), idx, w, m)         -> (B,C,m)
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libs2c_oracle.so")


def build(force=False):
    """Compile oracle/s2c_oracle.c with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "s2c_oracle.c")
    if (not force and os.path.exists(_SO)
            and os.path.getmtime(_SO) >= os.path.getmtime(src)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        # the same switch as the kernels' build-time -DS2C_NVCC_CONTRACT (csrc/s2c_common.h)
        _lib.s2c_oracle_set_contract(int(os.environ.get("S2C_NVCC_CONTRACT", "0") or 0))
    return _lib


def set_contract(mode):
    """0: canonical arithmetic; 1 | 2: a*a + b*b + c*c of the index-producing ops with the fused
    multiply-adds of an nvcc build (oracle/s2c_oracle.c: sq3).  Returns the previous mode."""
    old = int(lib().s2c_oracle_get_contract())
    lib().s2c_oracle_set_contract(int(mode))
    return old


_F = ctypes.POINTER(ctypes.c_float)
_I = ctypes.POINTER(ctypes.c_int)


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_F)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_I)


def opt_n_threads(work_size):
    return int(lib().s2c_oracle_opt_n_threads(int(work_size)))


def furthest_point_sampling(xyz, m):
    xyz, pxyz = _f(xyz)
    b, n, _ = xyz.shape
    temp = np.empty((b, n), np.float32)
    idx = np.zeros((b, m), np.int32)
    lib().s2c_oracle_furthest_point_sampling(
        b, n, int(m), pxyz, temp.ctypes.data_as(_F), idx.ctypes.data_as(_I))
    return idx


def gather_points(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().s2c_oracle_gather_points(b, c, n, m, pp, pi, out.ctypes.data_as(_F))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().s2c_oracle_gather_points_grad(b, c, int(n), m, pg, pi,
                                        out.ctypes.data_as(_F))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz, pq = _f(new_xyz)
    xyz, pp = _f(xyz)
    b, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    lib().s2c_oracle_ball_query(b, n, m, ctypes.c_float(radius), int(nsample),
                                pq, pp, idx.ctypes.data_as(_I))
    return idx


def group_points(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = np.zeros((b, c, npoints, nsample), np.float32)
    lib().s2c_oracle_group_points(b, c, n, npoints, nsample, pp, pi,
                                  out.ctypes.data_as(_F))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    b, c, npoints, nsample = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().s2c_oracle_group_points_grad(b, c, int(n), npoints, nsample, pg, pi,
                                       out.ctypes.data_as(_F))
    return out


def three_nn(unknown, known):
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    lib().s2c_oracle_three_nn(b, n, m, pu, pk, dist2.ctypes.data_as(_F),
                              idx.ctypes.data_as(_I))
    return dist2, idx


def three_interpolate(points, idx, weight):
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib().s2c_oracle_three_interpolate(b, c, m, n, pp, pi, pw,
                                       out.ctypes.data_as(_F))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, m), np.float32)
    lib().s2c_oracle_three_interpolate_grad(b, c, n, int(m), pg, pi, pw,
                                            out.ctypes.data_as(_F))
    return out
