"""`parse_predictions` (lib/ap_helper.py:40-178) with the box decode, the empty-box
test and the NMS on the device (csrc/s2c_post.hip).  Same signature, same return
value (`batch_pred_map_cls`: per scene a list of (class, corners (8,3) float64, score))
and the same `end_points` side effects (`pred_mask`, `batch_pred_map_cls`).

Reference cost: B*K scipy Delaunay hull tests over N points, a Python
double loop of `get_3d_box`, numpy NMS and ~10 D2H copies with `.item()`.  Here: one
device pass, ONE host synchronisation at the end (the result is a Python list).
"""
import ctypes
import math

import numpy as np
import torch

from . import _C
from .box_util import get_3d_box_batch

_I, _L, _D, _P = ctypes.c_int, ctypes.c_longlong, ctypes.c_double, ctypes.c_void_p
_C.register("s2c_boxes_count_points", [_I, _I, _I, _P, _L, _L, _P, _P, _P, _P, _P])
_C.register("s2c_nms", [_I, _I, _P, _P, _P, _P, _D, _I, _I, _P, _P])

MIN_POINTS_IN_BOX = 5      # ap_helper.py:101


def _heading_angle(cfg, heading_class, heading_residual):
    """class2angle for every proposal.  ScanNet boxes are axis aligned
    (model_util_scannet.py:131-135 returns 0); a config with real heading bins follows
    VoteNet's class2angle (class centre + residual, wrapped to (-pi, pi])."""
    nh = int(getattr(cfg, "num_heading_bin", 1))
    if getattr(cfg, "axis_aligned", nh == 1):
        return torch.zeros_like(heading_residual, dtype=torch.float64)
    ang = heading_class.double() * (2.0 * math.pi / nh) + heading_residual.double()
    return torch.where(ang > math.pi, ang - 2.0 * math.pi, ang)


def decode_boxes(end_points, cfg):
    """-> dict of device tensors: classes, float64 box parameters and corners."""
    center = end_points["center"]
    dev = center.device
    hcls = torch.argmax(end_points["heading_scores"], -1)
    hres = torch.gather(end_points["heading_residuals"], 2, hcls.unsqueeze(-1)).squeeze(2)
    scls = torch.argmax(end_points["size_scores"], -1)
    sres = torch.gather(end_points["size_residuals"], 2,
                        scls.view(*scls.shape, 1, 1).expand(-1, -1, 1, 3)).squeeze(2)
    sem = torch.argmax(end_points["sem_cls_scores"], -1)
    msa = torch.as_tensor(np.asarray(cfg.mean_size_arr, np.float64), device=dev)
    size = msa[scls] + sres.double()                      # class2size
    angle = _heading_angle(cfg, hcls, hres)
    c64 = center.double()
    corners = get_3d_box_batch(size, angle, c64)          # (B,K,8,3) float64
    return dict(center=c64.contiguous(), size=size.contiguous(), angle=angle.contiguous(),
                corners=corners, sem_cls=sem, heading_class=hcls, size_class=scls)


def nonempty_box_mask(point_clouds, boxes):
    """(B,K) bool: at least MIN_POINTS_IN_BOX points inside (ap_helper.py:92-103)."""
    B, K = boxes["angle"].shape
    pc = point_clouds
    if pc.stride(2) != 1:
        pc = pc.contiguous()
    counts = torch.empty((B, K), dtype=torch.int32, device=pc.device)
    with torch.cuda.device(pc.device):
        _C.call("s2c_boxes_count_points", B, pc.shape[1], K, pc.data_ptr(), pc.stride(1),
                pc.stride(0), boxes["center"].data_ptr(), boxes["size"].data_ptr(),
                boxes["angle"].data_ptr(), counts.data_ptr(), _C.stream_ptr())
    return counts >= MIN_POINTS_IN_BOX


def nms_mask(boxes, obj_prob, nonempty, config_dict):
    """(B,K) uint8 keep mask of the configured NMS flavour (ap_helper.py:107-161)."""
    corners = boxes["corners"]
    B, K = obj_prob.shape
    lo, hi = corners.min(2)[0], corners.max(2)[0]         # (B,K,3)
    cls = None
    add_eps = 0
    if not config_dict["use_3d_nms"]:
        # 2-D boxes on the x / z extents (:112-116); unit third extent
        bx = torch.stack([lo[..., 0], lo[..., 2], torch.zeros_like(lo[..., 0]),
                          hi[..., 0], hi[..., 2], torch.ones_like(lo[..., 0])], -1)
    else:
        bx = torch.cat([lo, hi], -1)
        if config_dict.get("cls_nms", False):
            cls = boxes["sem_cls"].to(torch.int64).contiguous()
            add_eps = 1
    bx = bx.contiguous()
    score = obj_prob.double().contiguous()
    valid = nonempty.to(torch.uint8).contiguous()
    keep = torch.empty((B, K), dtype=torch.uint8, device=bx.device)
    with torch.cuda.device(bx.device):
        _C.call("s2c_nms", B, K, bx.data_ptr(), score.data_ptr(),
                cls.data_ptr() if cls is not None else None, valid.data_ptr(),
                float(config_dict["nms_iou"]), int(bool(config_dict["use_old_type_nms"])),
                add_eps, keep.data_ptr(), _C.stream_ptr())
    return keep


def parse_predictions(end_points, config_dict):
    """lib/ap_helper.py:40-178."""
    cfg = config_dict["dataset_config"]
    if not end_points["center"].is_cuda:
        raise RuntimeError("scan2cap_amd.ap_helper.parse_predictions: CPU not supported "
                           "(the post-processing kernels are HIP-only)")
    boxes = decode_boxes(end_points, cfg)
    B, K = boxes["angle"].shape
    sem_probs = torch.softmax(end_points["sem_cls_scores"].float(), -1)
    obj_prob = torch.softmax(end_points["objectness_scores"].float(), -1)[:, :, 1]
    if config_dict["remove_empty_box"]:
        nonempty = nonempty_box_mask(end_points["point_clouds"], boxes)
    else:
        nonempty = torch.ones((B, K), dtype=torch.bool, device=obj_prob.device)
    keep = nms_mask(boxes, obj_prob, nonempty, config_dict)
    end_points["pred_mask_device"] = keep
    # ---- the only host synchronisation: the API returns Python lists --------------
    keep_h = keep.cpu().numpy()
    if not (nonempty.any(1)).all().item():
        raise AssertionError("parse_predictions: a scene has no non-empty box "
                             "(ap_helper.py:121 `assert(len(pick)>0)`)")
    corners_h = boxes["corners"].cpu().numpy()
    obj_h = obj_prob.cpu().numpy()
    sem_h = boxes["sem_cls"].cpu().numpy()
    semp_h = sem_probs.cpu().numpy()
    end_points["pred_mask"] = keep_h.astype(np.float64)
    conf = config_dict["conf_thresh"]
    out = []
    for i in range(B):
        sel = [j for j in range(K) if keep_h[i, j] == 1 and obj_h[i, j] > conf]
        if config_dict["per_class_proposal"]:
            cur = []
            for ii in range(cfg.num_class):
                cur += [(ii, corners_h[i, j], semp_h[i, j, ii] * obj_h[i, j]) for j in sel]
        else:
            cur = [(int(sem_h[i, j]), corners_h[i, j], obj_h[i, j]) for j in sel]
        out.append(cur)
    end_points["batch_pred_map_cls"] = out
    return out
