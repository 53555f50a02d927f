"""`torch.ops.s2c.*` -- the nine point-cloud operators registered as PyTorch custom ops.

The reference exposes them through a torch extension module (`pointnet2._ext`,
lib/pointnet2/_ext_src/src/bindings.cpp:6-19: nine `m.def`s) and wraps each in a
hand-written `autograd.Function` (lib/pointnet2/pointnet2_utils.py).  Here the same nine
entry points are registered with `torch.library` on top of the C ABI of libs2c_hip.so
(include/s2c_ops.h):

* schema   `s2c::<name>(...)`, the argument order of the C++ wrappers (note ball_query:
           new_xyz, xyz, radius, nsample);
* kernel   the "CUDA" dispatch key (HIP on ROCm) -> scan2cap_amd.pointnet2._ext (ctypes ->
           C ABI, current HIP stream, no synchronisation).  There is NO CPU kernel: a CPU
           tensor raises NotImplementedError from the dispatcher, as the reference's
           AT_ASSERT("CPU not supported") does;
* fake     shape / dtype inference for meta tensors, `torch.compile` tracing, FakeTensorMode
           and `torch.library.opcheck`;
* autograd gather_points / group_points / three_interpolate differentiate through their
           `*_grad` ops (sampling_gpu.cu:34-57, group_points_gpu.cu:43-75,
           interpolate_gpu.cu:108-154); FPS / ball_query / three_nn outputs are
           non-differentiable (pointnet2_utils.py:74, :143, :285).

`scan2cap_amd.pointnet2.pointnet2_utils` routes its autograd wrappers through these ops.
"""
import torch
from torch.library import Library, impl, register_autograd, register_fake

from . import _ext

NAMES = ("furthest_point_sampling", "gather_points", "gather_points_grad", "ball_query",
         "group_points", "group_points_grad", "three_nn", "three_interpolate",
         "three_interpolate_grad")

_lib = Library("s2c", "DEF")
_lib.define("furthest_point_sampling(Tensor xyz, int npoint) -> Tensor")
_lib.define("gather_points(Tensor points, Tensor idx) -> Tensor")
_lib.define("gather_points_grad(Tensor grad_out, Tensor idx, int n) -> Tensor")
_lib.define("ball_query(Tensor new_xyz, Tensor xyz, float radius, int nsample) -> Tensor")
_lib.define("group_points(Tensor points, Tensor idx) -> Tensor")
_lib.define("group_points_grad(Tensor grad_out, Tensor idx, int n) -> Tensor")
_lib.define("three_nn(Tensor unknown, Tensor known) -> (Tensor, Tensor)")
_lib.define("three_interpolate(Tensor points, Tensor idx, Tensor weight) -> Tensor")
_lib.define("three_interpolate_grad(Tensor grad_out, Tensor idx, Tensor weight, int m) -> Tensor")


# ---- device kernels (dispatch key "CUDA" = HIP on ROCm) -----------------------------------
@impl(_lib, "furthest_point_sampling", "CUDA")
def _fps(xyz, npoint):
    return _ext.furthest_point_sampling(xyz, npoint)


@impl(_lib, "gather_points", "CUDA")
def _gather(points, idx):
    return _ext.gather_points(points, idx)


@impl(_lib, "gather_points_grad", "CUDA")
def _gather_grad(grad_out, idx, n):
    return _ext.gather_points_grad(grad_out, idx, n)


@impl(_lib, "ball_query", "CUDA")
def _ball_query(new_xyz, xyz, radius, nsample):
    return _ext.ball_query(new_xyz, xyz, radius, nsample)


@impl(_lib, "group_points", "CUDA")
def _group(points, idx):
    return _ext.group_points(points, idx)


@impl(_lib, "group_points_grad", "CUDA")
def _group_grad(grad_out, idx, n):
    return _ext.group_points_grad(grad_out, idx, n)


@impl(_lib, "three_nn", "CUDA")
def _three_nn(unknown, known):
    d2, idx = _ext.three_nn(unknown, known)
    return d2, idx


@impl(_lib, "three_interpolate", "CUDA")
def _interp(points, idx, weight):
    return _ext.three_interpolate(points, idx, weight)


@impl(_lib, "three_interpolate_grad", "CUDA")
def _interp_grad(grad_out, idx, weight, m):
    return _ext.three_interpolate_grad(grad_out, idx, weight, m)


# ---- fake (meta) implementations -------------------------------------------------------------
def _f32(x, *shape):
    return x.new_empty(shape, dtype=torch.float32)


def _i32(x, *shape):
    return x.new_empty(shape, dtype=torch.int32)


@register_fake("s2c::furthest_point_sampling")
def _(xyz, npoint):
    return _i32(xyz, xyz.shape[0], npoint)


@register_fake("s2c::gather_points")
def _(points, idx):
    return _f32(points, points.shape[0], points.shape[1], idx.shape[1])


@register_fake("s2c::gather_points_grad")
def _(grad_out, idx, n):
    return _f32(grad_out, grad_out.shape[0], grad_out.shape[1], n)


@register_fake("s2c::ball_query")
def _(new_xyz, xyz, radius, nsample):
    return _i32(new_xyz, new_xyz.shape[0], new_xyz.shape[1], nsample)


@register_fake("s2c::group_points")
def _(points, idx):
    return _f32(points, points.shape[0], points.shape[1], idx.shape[1], idx.shape[2])


@register_fake("s2c::group_points_grad")
def _(grad_out, idx, n):
    return _f32(grad_out, grad_out.shape[0], grad_out.shape[1], n)


@register_fake("s2c::three_nn")
def _(unknown, known):
    b, n = unknown.shape[0], unknown.shape[1]
    return _f32(unknown, b, n, 3), _i32(unknown, b, n, 3)


@register_fake("s2c::three_interpolate")
def _(points, idx, weight):
    return _f32(points, points.shape[0], points.shape[1], idx.shape[1])


@register_fake("s2c::three_interpolate_grad")
def _(grad_out, idx, weight, m):
    return _f32(grad_out, grad_out.shape[0], grad_out.shape[1], m)


# ---- autograd ------------------------------------------------------------------------------
def _gather_setup(ctx, inputs, output):
    points, idx = inputs
    ctx.n = points.shape[2]
    ctx.save_for_backward(idx)


def _gather_bwd(ctx, grad_out):
    (idx,) = ctx.saved_tensors
    return torch.ops.s2c.gather_points_grad(grad_out.contiguous(), idx, ctx.n), None


register_autograd("s2c::gather_points", _gather_bwd, setup_context=_gather_setup)


def _group_setup(ctx, inputs, output):
    points, idx = inputs
    ctx.n = points.shape[2]
    ctx.save_for_backward(idx)


def _group_bwd(ctx, grad_out):
    (idx,) = ctx.saved_tensors
    return torch.ops.s2c.group_points_grad(grad_out.contiguous(), idx, ctx.n), None


register_autograd("s2c::group_points", _group_bwd, setup_context=_group_setup)


def _interp_setup(ctx, inputs, output):
    points, idx, weight = inputs
    ctx.m = points.shape[2]
    ctx.save_for_backward(idx, weight)


def _interp_bwd(ctx, grad_out):
    idx, weight = ctx.saved_tensors
    return (torch.ops.s2c.three_interpolate_grad(grad_out.contiguous(), idx, weight, ctx.m),
            None, None)


register_autograd("s2c::three_interpolate", _interp_bwd, setup_context=_interp_setup)
