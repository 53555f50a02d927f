"""Axis-aligned box helpers that stay on the device.

The reference decodes boxes in numpy float64 on the CPU
(models/proposal_module.py:80-103 -> DC.param2obb_batch,
data/scannet/model_util_scannet.py:165-172 -> utils/box_util.py:360-383), a
device->host->device round trip in the middle of forward.  The same float64
arithmetic is done here with torch ops on the GPU; results are bit-identical for
heading 0 (ScanNet boxes are axis aligned: class2angle_batch returns zeros,
model_util_scannet.py:142-146).
"""
import math

import torch

from .consts import const

# corner sign pattern of utils/box_util.py:377-379 (x: l, y: w, z: h)
_SX = (1, 1, -1, -1, 1, 1, -1, -1)
_SY = (1, -1, -1, 1, 1, -1, -1, 1)
_SZ = (1, 1, 1, 1, -1, -1, -1, -1)


def get_3d_box_batch(box_size, heading_angle, center):
    """box_size (...,3), heading_angle (...), center (...,3) -> (...,8,3).

    Follows utils/box_util.py:360-383 including its rotation matrix
    (roty_batch, :323-339) so non-zero headings stay supported; dtype follows
    the inputs (the reference runs it in float64)."""
    l = box_size[..., 0:1]
    w = box_size[..., 1:2]
    h = box_size[..., 2:3]
    sx = const("box_sx", _SX, box_size.device, box_size.dtype)
    sy = const("box_sy", _SY, box_size.device, box_size.dtype)
    sz = const("box_sz", _SZ, box_size.device, box_size.dtype)
    cx = (l / 2) * sx
    cy = (w / 2) * sy
    cz = (h / 2) * sz
    corners = torch.stack([cx, cy, cz], -1)  # (...,8,3)
    c = torch.cos(heading_angle)
    s = torch.sin(heading_angle)
    zeros = torch.zeros_like(c)
    ones = torch.ones_like(c)
    # roty_batch (box_util.py:323-339): R = [[c,0,s],[0,1,0],[-s,0,c]];
    # corners @ R^T  (box_util.py:380-381)
    R = torch.stack([torch.stack([c, zeros, s], -1),
                     torch.stack([zeros, ones, zeros], -1),
                     torch.stack([-s, zeros, c], -1)], -2)
    corners = torch.matmul(corners, R.transpose(-1, -2))
    return corners + center.unsqueeze(-2)


def box_min_max(corners):
    """(...,8,3) -> min (...,3), max (...,3)   (box_util.py:211-233)."""
    return corners.min(dim=-2)[0], corners.max(dim=-2)[0]


def box3d_iou_batch_tensor(corners1, corners2):
    """AABB IoU, utils/box_util.py:183-209.  (...,8,3) x (...,8,3) -> (...)
    with broadcasting over the leading dims."""
    min1, max1 = box_min_max(corners1)
    min2, max2 = box_min_max(corners2)
    return aabb_iou(min1, max1, min2, max2)


def aabb_iou(min1, max1, min2, max2):
    lo = torch.max(min1, min2)
    hi = torch.min(max1, max2)
    d = torch.clamp(hi - lo, min=0)
    inter = d[..., 0] * d[..., 1] * d[..., 2]
    e1 = max1 - min1
    e2 = max2 - min2
    vol1 = e1[..., 0] * e1[..., 1] * e1[..., 2]
    vol2 = e2[..., 0] * e2[..., 1] * e2[..., 2]
    return inter / (vol1 + vol2 - inter + 1e-8)


__all__ = ["get_3d_box_batch", "box3d_iou_batch_tensor", "aabb_iou",
           "box_min_max", "math"]
