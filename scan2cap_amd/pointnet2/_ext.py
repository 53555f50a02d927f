"""Drop-in replacement of the reference's pybind module `pointnet2._ext`
(lib/pointnet2/_ext_src/src/bindings.cpp:6-19): the same nine functions, same
argument order, same tensor layouts and dtypes, same error class (the
reference's AT_ASSERT checks surface as RuntimeError, include/utils.h:5-25).

Outputs are fresh tensors on the inputs' device; inputs are borrowed.  All work
is enqueued on the current HIP stream, nothing synchronises.  CPU tensors are
rejected exactly like the reference ("CPU not supported",
ball_query.cpp:27-29) -- there is no fallback.
"""
import os

import torch

from .. import _C


def _check(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def _chk_f(x, name):
    _check(x.is_contiguous(), "%s must be a contiguous tensor" % name)
    _check(x.dtype == torch.float32, "%s must be a float tensor" % name)
    _check(x.is_cuda, "CPU not supported")


def _chk_i(x, name):
    _check(x.is_contiguous(), "%s must be a contiguous tensor" % name)
    _check(x.dtype == torch.int32, "%s must be an int tensor" % name)
    _check(x.is_cuda, "%s must be a CUDA tensor" % name)


def _run(name, ref, *args, alg_bytes=0, allow=()):
    """alg_bytes: algorithmic HBM bytes of this call under the op contract
    (compulsory reads + writes), recorded only while bench.py's
    kernel timer is on.  allow: return codes handed back instead of raised."""
    if _C.TIMER.enabled:
        _C.TIMER.alg_bytes = int(alg_bytes)
    with torch.cuda.device(ref.device):
        return _C.call(name, *args, _C.stream_ptr(), allow=allow)


def gather_points(points, idx):
    """sampling.cpp:15-38.  (B,C,N) f32, (B,m) i32 -> (B,C,m)"""
    _chk_f(points, "points")
    _chk_i(idx, "idx")
    b, c, n = points.shape
    m = idx.shape[1]
    out = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    if out.numel() == 0:
        return out
    _run("s2c_gather_points", points, b, c, n, m, points.data_ptr(),
         idx.data_ptr(), out.data_ptr(), alg_bytes=4 * (2 * b * c * m + b * m))
    return out


def gather_points_grad(grad_out, idx, n):
    """sampling.cpp:40-65.  (B,C,m), (B,m), n -> (B,C,n)"""
    _chk_f(grad_out, "grad_out")
    _chk_i(idx, "idx")
    b, c, m = grad_out.shape
    out = torch.empty((b, c, n), dtype=torch.float32, device=grad_out.device)
    if out.numel() == 0 or m == 0:
        return out.zero_()
    _run("s2c_gather_points_grad", grad_out, b, c, int(n), m,
         grad_out.data_ptr(), idx.data_ptr(), out.data_ptr(),
         alg_bytes=4 * (b * c * m + b * m + b * c * n))
    return out


# point sets at least this large take the spatially-bucketed exact kernel
# (csrc/s2c_fps_bucket.hip); smaller ones stay register-resident (csrc/s2c_ops.hip)
FPS_BUCKET_MIN_N = 8192


def furthest_point_sampling(points, nsamples, prefix_hint=False, return_fallback=False):
    """sampling.cpp:66-87.  (B,N,3) f32 -> (B,nsamples) i32
    prefix_hint: the caller expects `points` to be in FPS pick order already (the centres of a
    previous FPS): the picks 0..nsamples-1 are then PROVEN per scene by a parallel kernel pair
    and the serial rounds run only for scenes where the proof fails -- same result for every
    input (csrc/s2c_fps_small.hip).  return_fallback: also return the (B,) int32 flags
    (1 = the scene ran the rounds)."""
    _chk_f(points, "points")
    b, n, _ = points.shape
    out = torch.empty((b, nsamples), dtype=torch.int32, device=points.device)
    if out.numel() == 0:               # empty batch / no samples asked for: nothing to launch
        if return_fallback:
            return out, torch.zeros(b, dtype=torch.int32, device=points.device)
        return out
    _check(n > 0, "furthest_point_sampling: no points to sample from")
    ab = 4 * (3 * b * n + b * nsamples)
    if prefix_hint and n <= _C.load().s2c_fps_small_limit() and FPS_PREFIX_VERIFY:
        ws = torch.empty(_C.load().s2c_fps_prefix_workspace_bytes(b, int(nsamples)),
                         dtype=torch.uint8, device=points.device)
        _run("s2c_furthest_point_sampling_prefix", points, b, n, int(nsamples),
             points.data_ptr(), ws.data_ptr(), out.data_ptr(), int(FPS_SMALL_THREADS),
             alg_bytes=ab)
        if return_fallback:
            return out, ws[:4 * b].view(torch.int32)
        return out
    if return_fallback:
        raise ValueError("return_fallback needs prefix_hint and n <= s2c_fps_small_limit()")
    if n >= FPS_BUCKET_MIN_N and FPS_LARGE_IMPL == "cells":
        # wave-owned grid cells, one barrier per round (csrc/s2c_fps_cells.hip)
        ws = torch.empty(_C.load().s2c_fps_cells_workspace_bytes(b, n), dtype=torch.uint8,
                         device=points.device)
        _run("s2c_furthest_point_sampling_cells", points, b, n, int(nsamples),
             points.data_ptr(), ws.data_ptr(), out.data_ptr(), int(FPS_CELLS_WAVES),
             alg_bytes=ab)
        return out
    if n >= FPS_BUCKET_MIN_N:
        ws = torch.empty(_C.load().s2c_fps_workspace_bytes(b, n), dtype=torch.uint8,
                         device=points.device)
        _run("s2c_furthest_point_sampling_bucketed", points, b, n, int(nsamples),
             points.data_ptr(), ws.data_ptr(), out.data_ptr(), alg_bytes=ab)
        return out
    _run("s2c_furthest_point_sampling_small", points, b, n, int(nsamples),
         points.data_ptr(), out.data_ptr(), int(FPS_SMALL_THREADS), alg_bytes=ab)
    return out


FPS_SMALL_THREADS = 0   # 0 = library heuristic (tests sweep 64..1024)
FPS_PREFIX_VERIFY = True   # honour prefix_hint
FPS_LARGE_IMPL = "cells"   # "cells" (s2c_fps_cells.hip) | "bucket" (s2c_fps_bucket.hip)
FPS_CELLS_WAVES = 0        # rounds-kernel workgroup in waves: 4 / 8 / 16, 0 = library default


def furthest_point_sampling_bruteforce(points, nsamples):
    """The register-resident / streaming brute-force kernel regardless of size
    (kept callable for A/B parity tests against the bucketed kernel)."""
    _chk_f(points, "points")
    b, n, _ = points.shape
    out = torch.empty((b, nsamples), dtype=torch.int32, device=points.device)
    temp = torch.empty((b, n), dtype=torch.float32, device=points.device)
    _run("s2c_furthest_point_sampling", points, b, n, int(nsamples),
         points.data_ptr(), temp.data_ptr(), out.data_ptr())
    return out


def three_nn(unknowns, knows):
    """interpolate.cpp:14-40.  (B,n,3), (B,m,3) -> [dist2 (B,n,3) f32, idx (B,n,3) i32]"""
    _chk_f(unknowns, "unknowns")
    _chk_f(knows, "knows")
    b, n, _ = unknowns.shape
    m = knows.shape[1]
    dist2 = torch.empty((b, n, 3), dtype=torch.float32, device=unknowns.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=unknowns.device)
    if idx.numel() == 0:
        return [dist2, idx]
    _run("s2c_three_nn", unknowns, b, n, m, unknowns.data_ptr(),
         knows.data_ptr(), dist2.data_ptr(), idx.data_ptr(),
         alg_bytes=4 * (3 * b * n + 3 * b * m + 6 * b * n))
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """interpolate.cpp:42-70.  (B,C,m), (B,n,3) i32, (B,n,3) f32 -> (B,C,n)"""
    _chk_f(points, "points")
    _chk_i(idx, "idx")
    _chk_f(weight, "weight")
    b, c, m = points.shape
    n = idx.shape[1]
    out = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
    if out.numel() == 0:
        return out
    _run("s2c_three_interpolate", points, b, c, m, n, points.data_ptr(),
         idx.data_ptr(), weight.data_ptr(), out.data_ptr(),
         alg_bytes=4 * (min(b * c * m, 3 * b * c * n) + 6 * b * n + b * c * n))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """interpolate.cpp:71-99.  (B,C,n), (B,n,3), (B,n,3), m -> (B,C,m)"""
    _chk_f(grad_out, "grad_out")
    _chk_i(idx, "idx")
    _chk_f(weight, "weight")
    b, c, n = grad_out.shape
    out = torch.empty((b, c, m), dtype=torch.float32, device=grad_out.device)
    if out.numel() == 0 or n == 0:
        return out.zero_()
    _run("s2c_three_interpolate_grad", grad_out, b, c, n, int(m),
         grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(),
         out.data_ptr(), alg_bytes=4 * (b * c * n + 6 * b * n + b * c * m))
    return out


# point sets at least this large take the grid kernel; smaller ones the LDS-tiled
# brute-force kernel (csrc/s2c_ops.hip), which is faster there
BQ_GRID_MIN_N = 4096


def ball_query_bruteforce(new_xyz, xyz, radius, nsample):
    """The brute-force kernel regardless of size (A/B parity tests)."""
    _chk_f(new_xyz, "new_xyz")
    _chk_f(xyz, "xyz")
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=new_xyz.device)
    _run("s2c_ball_query", new_xyz, b, n, m, float(radius), int(nsample),
         new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr())
    return idx


def ball_query(new_xyz, xyz, radius, nsample):
    """ball_query.cpp:8-32.  NOTE the C++ argument order (new_xyz, xyz, radius,
    nsample) differs from the Python wrapper's (pointnet2_utils.py:262,282)."""
    _chk_f(new_xyz, "new_xyz")
    _chk_f(xyz, "xyz")
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=new_xyz.device)
    if idx.numel() == 0 or n == 0:     # nothing to find: rows stay 0 (ball_query.cpp:19-21)
        return idx.zero_()
    ab = 4 * (3 * b * n + 3 * b * m + b * m * nsample)
    if n >= BQ_GRID_MIN_N and 0 < nsample <= 64 and radius > 0 and b > 0 and m > 0:
        # large clouds: uniform grid, a centre visits <= 27 cells (csrc/s2c_bq_grid.hip)
        ws = torch.empty(_C.load().s2c_ball_query_workspace_bytes(b, n), dtype=torch.uint8,
                         device=new_xyz.device)
        rc = _run("s2c_ball_query_grid", new_xyz, b, n, m, float(radius), int(nsample),
                  new_xyz.data_ptr(), xyz.data_ptr(), ws.data_ptr(), idx.data_ptr(),
                  alg_bytes=ab, allow=(-2,))
        if rc == 0:
            return idx
        # S2C_ENOSUP: this device refused the build kernel's LDS size -> brute-force kernel
    _run("s2c_ball_query", new_xyz, b, n, m, float(radius), int(nsample),
         new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr(),
         alg_bytes=4 * (3 * b * n + 3 * b * m + b * m * nsample))
    return idx


def group_points(points, idx):
    """group_points.cpp:12-36.  (B,C,N), (B,m,ns) i32 -> (B,C,m,ns)"""
    _chk_f(points, "points")
    _chk_i(idx, "idx")
    b, c, n = points.shape
    _, m, ns = idx.shape
    out = torch.empty((b, c, m, ns), dtype=torch.float32, device=points.device)
    if out.numel() == 0:
        return out
    _run("s2c_group_points", points, b, c, n, m, ns, points.data_ptr(),
         idx.data_ptr(), out.data_ptr(),
         alg_bytes=4 * (min(b * c * n, b * c * m * ns) + b * m * ns + b * c * m * ns))
    return out


def group_points_grad(grad_out, idx, n):
    """group_points.cpp:38-62.  (B,C,m,ns), (B,m,ns), n -> (B,C,n)"""
    _chk_f(grad_out, "grad_out")
    _chk_i(idx, "idx")
    b, c, m, ns = grad_out.shape
    out = torch.empty((b, c, n), dtype=torch.float32, device=grad_out.device)
    if out.numel() == 0 or m * ns == 0:
        return out.zero_()
    _run("s2c_group_points_grad", grad_out, b, c, int(n), m, ns,
         grad_out.data_ptr(), idx.data_ptr(), out.data_ptr(),
         alg_bytes=4 * (b * c * m * ns + b * m * ns + b * c * n))
    return out
