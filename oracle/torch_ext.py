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


def gather_points(points, idx):
    _chk(points); _chk(idx)
    return _t(orc.gather_points(points.detach().numpy(), idx.numpy()))


def gather_points_grad(grad_out, idx, n):
    return _t(orc.gather_points_grad(grad_out.detach().numpy(), idx.numpy(), n))


def furthest_point_sampling(points, nsamples):
    _chk(points)
    return _t(orc.furthest_point_sampling(points.detach().numpy(), nsamples))


def three_nn(unknowns, knows):
    d, i = orc.three_nn(unknowns.detach().numpy(), knows.detach().numpy())
    return [_t(d), _t(i)]


def three_interpolate(points, idx, weight):
    return _t(orc.three_interpolate(points.detach().numpy(), idx.numpy(),
                                    weight.detach().numpy()))


def three_interpolate_grad(grad_out, idx, weight, m):
    return _t(orc.three_interpolate_grad(grad_out.detach().numpy(), idx.numpy(),
                                         weight.detach().numpy(), m))


def ball_query(new_xyz, xyz, radius, nsample):
    _chk(new_xyz); _chk(xyz)
    return _t(orc.ball_query(new_xyz.detach().numpy(), xyz.detach().numpy(),
                             radius, nsample))


def group_points(points, idx):
    _chk(points); _chk(idx)
    return _t(orc.group_points(points.detach().numpy(), idx.numpy()))


def group_points_grad(grad_out, idx, n):
    return _t(orc.group_points_grad(grad_out.detach().numpy(), idx.numpy(), n))


NAMES = ("gather_points", "gather_points_grad", "furthest_point_sampling",
         "three_nn", "three_interpolate", "three_interpolate_grad",
         "ball_query", "group_points", "group_points_grad")


def as_module(name="pointnet2._ext"):
    mod = types.ModuleType(name)
    for n in NAMES:
        setattr(mod, n, globals()[n])
    return mod
