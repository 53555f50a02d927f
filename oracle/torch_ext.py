"""The CPU oracle dressed as a `pointnet2._ext` module for CPU torch tensors.

TEST INFRASTRUCTURE ONLY.  Used (a) by oracle/ref_harness.py to run the
reference's Python layers in this container and (b) by the CPU tests as a test
double injected into scan2cap_amd.pointnet2._ext so the build's host logic can
be checked against the golden fixtures without a GPU.  The product never
imports it.
"""
import types

import numpy as np
import torch

from . import oracle as orc


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _chk(x):
    if not x.is_contiguous():
        raise RuntimeError("must be a contiguous tensor")


# float64 tensors (tests/gen_golden.py runs the reference once more in double as the TRUTH its own
# float32 outputs and the HIP path are measured against): the index-producing ops decide on the
# float32 cast of their coordinates -- the reference's own discrete decisions, the C restatement --
# the float ops (pure gathers / weighted sums / scatter-adds) run in numpy float64.
def _is64(*ts):
    return any(t.dtype == torch.float64 for t in ts)


def _f32(t):
    return np.ascontiguousarray(t.detach().numpy().astype(np.float32))


def gather_points(points, idx):
    _chk(points); _chk(idx)
    if _is64(points):
        p, i = points.detach().numpy(), idx.numpy().astype(np.int64)
        return _t(np.take_along_axis(p, i[:, None, :], axis=2))
    return _t(orc.gather_points(points.detach().numpy(), idx.numpy()))


def gather_points_grad(grad_out, idx, n):
    if _is64(grad_out):
        g, i = grad_out.detach().numpy(), idx.numpy().astype(np.int64)
        B, C, m = g.shape
        out = np.zeros((B, C, n), np.float64)
        for b in range(B):
            np.add.at(out[b], (slice(None), i[b]), g[b])
        return _t(out)
    return _t(orc.gather_points_grad(grad_out.detach().numpy(), idx.numpy(), n))


def furthest_point_sampling(points, nsamples):
    _chk(points)
    if _is64(points):
        return _t(orc.furthest_point_sampling(_f32(points), nsamples))
    return _t(orc.furthest_point_sampling(points.detach().numpy(), nsamples))


def three_nn(unknowns, knows):
    if _is64(unknowns, knows):
        _, i = orc.three_nn(_f32(unknowns), _f32(knows))
        u, k = unknowns.detach().numpy(), knows.detach().numpy()
        sel = np.take_along_axis(k[:, None, :, :], i.astype(np.int64)[..., None], axis=2)   # (B,n,3,3)
        d = ((u[:, :, None, :] - sel) ** 2).sum(-1)          # squared, as the op returns it
        return [_t(d), _t(i)]
    d, i = orc.three_nn(unknowns.detach().numpy(), knows.detach().numpy())
    return [_t(d), _t(i)]


def three_interpolate(points, idx, weight):
    if _is64(points, weight):
        p, i, w = points.detach().numpy(), idx.numpy().astype(np.int64), weight.detach().numpy()
        B, C, m = p.shape
        n = i.shape[1]
        g = np.take_along_axis(p[:, :, None, :], i.reshape(B, 1, n * 3)[:, :, None, :].repeat(C, 1)
                               .reshape(B, C, 1, n * 3), axis=3).reshape(B, C, n, 3)
        return _t((g * w[:, None, :, :]).sum(-1))
    return _t(orc.three_interpolate(points.detach().numpy(), idx.numpy(),
                                    weight.detach().numpy()))


def three_interpolate_grad(grad_out, idx, weight, m):
    if _is64(grad_out, weight):
        g, i, w = grad_out.detach().numpy(), idx.numpy().astype(np.int64), weight.detach().numpy()
        B, C, n = g.shape
        out = np.zeros((B, C, m), np.float64)
        for b in range(B):
            for k in range(3):
                np.add.at(out[b], (slice(None), i[b, :, k]), g[b] * w[b, :, k][None, :])
        return _t(out)
    return _t(orc.three_interpolate_grad(grad_out.detach().numpy(), idx.numpy(),
                                         weight.detach().numpy(), m))


def ball_query(new_xyz, xyz, radius, nsample):
    _chk(new_xyz); _chk(xyz)
    if _is64(new_xyz, xyz):
        return _t(orc.ball_query(_f32(new_xyz), _f32(xyz), radius, nsample))
    return _t(orc.ball_query(new_xyz.detach().numpy(), xyz.detach().numpy(),
                             radius, nsample))


def group_points(points, idx):
    _chk(points); _chk(idx)
    if _is64(points):
        p, i = points.detach().numpy(), idx.numpy().astype(np.int64)
        B, C, n = p.shape
        _, m, ns = i.shape
        return _t(np.take_along_axis(p, i.reshape(B, 1, m * ns), axis=2).reshape(B, C, m, ns))
    return _t(orc.group_points(points.detach().numpy(), idx.numpy()))


def group_points_grad(grad_out, idx, n):
    if _is64(grad_out):
        g, i = grad_out.detach().numpy(), idx.numpy().astype(np.int64)
        B, C, m, ns = g.shape
        out = np.zeros((B, C, n), np.float64)
        for b in range(B):
            np.add.at(out[b], (slice(None), i[b].reshape(-1)), g[b].reshape(C, m * ns))
        return _t(out)
    return _t(orc.group_points_grad(grad_out.detach().numpy(), idx.numpy(), n))


NAMES = ("gather_points", "gather_points_grad", "furthest_point_sampling",
         "three_nn", "three_interpolate", "three_interpolate_grad",
         "ball_query", "group_points", "group_points_grad")


def as_module(name="pointnet2._ext"):
    mod = types.ModuleType(name)
    for n in NAMES:
        setattr(mod, n, globals()[n])
    return mod
