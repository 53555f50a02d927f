"""Detection part of `get_scene_cap_loss` (lib/loss_helper.py:24-187, :381-491) as
ONE autograd Function over csrc/s2c_loss.hip: 2 forward + 1 backward launches
instead of ~170 + ~250 micro-kernels.  Same terms, weights and
reductions; tie rules = torch.min / torch.argmax (first extremum).

Only the weighted detection total `10 * (vote + 0.5 objectness + box + 0.1 sem_cls)`
carries a gradient; the individual terms are returned detached (the reference only
ever logs them, lib/solver.py:314-330).
"""
import ctypes

import numpy as np
import torch
from torch.autograd import Function

from . import _C
from .consts import array_const

_I, _F, _P = ctypes.c_int, ctypes.c_float, ctypes.c_void_p

_INTS = ("B", "S", "VF", "N", "K", "G", "NH", "NS", "NC", "ld_center_label", "ld_scores")
_FLOATS = ("near_threshold", "far_threshold", "obj_w0", "obj_w1")
_PTRS = ("seed_xyz", "vote_xyz", "seed_inds", "vote_label", "vote_label_mask", "agg_xyz",
         "center_label", "objectness_scores", "center", "box_label_mask",
         "heading_class_label", "heading_residual_label", "size_class_label",
         "size_residual_label", "sem_cls_label", "heading_scores", "heading_res_norm",
         "size_scores", "size_res_norm", "sem_cls_scores", "mean_size_arr",
         "objectness_label", "objectness_mask", "object_assignment", "vote_arg",
         "center_g1", "center_k2", "partial", "stats")
_GRADS = ("vote_xyz", "objectness_scores", "center", "heading_scores", "heading_res_norm",
          "size_scores", "size_res_norm", "sem_cls_scores")


class _Args(ctypes.Structure):
    """include/s2c_fused.h: s2c_detloss_args"""
    _fields_ = ([(n, _I) for n in _INTS] + [(n, _F) for n in _FLOATS] +
                [(n, _P) for n in _PTRS])


class _Grads(ctypes.Structure):
    """include/s2c_fused.h: s2c_detloss_grads"""
    _fields_ = [(n, _P) for n in _GRADS]


_C.register("s2c_detection_loss_fwd", [_P, _P])
_C.register("s2c_detection_loss_bwd", [_P, _P, _P, _P])

STAT_NAMES = ("vote_loss", "objectness_loss", "center_loss", "heading_cls_loss",
              "heading_reg_loss", "size_cls_loss", "size_reg_loss", "sem_cls_loss",
              "box_loss")


def _c(t, dtype):
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


class DetectionLoss(Function):
    """Dense inputs: eight tensors in, eight gradients out.  Head-rows mode (`rows` = the
    proposal head's (B,K,nout) output whose column groups the score tensors are views of):
    the kernels read the scores in place (row stride nout) and write ONE gradient tensor
    d rows -- no contiguous copies of six slices, no slice-gradient assembly."""

    @staticmethod
    def forward(ctx, vote_xyz, objectness_scores, center, heading_scores, heading_res_norm,
                size_scores, size_res_norm, sem_cls_scores, labels, cfg, rows=None,
                agg_xyz=None):
        dev = center.device
        f32, i64 = torch.float32, torch.int64
        keep = (lambda x, dt: x) if rows is not None else _c
        t = {
            "vote_xyz": _c(vote_xyz, f32), "objectness_scores": keep(objectness_scores, f32),
            "center": _c(center, f32), "heading_scores": keep(heading_scores, f32),
            "heading_res_norm": keep(heading_res_norm, f32),
            "size_scores": keep(size_scores, f32), "size_res_norm": keep(size_res_norm, f32),
            "sem_cls_scores": keep(sem_cls_scores, f32),
            "seed_xyz": _c(labels["seed_xyz"], f32),
            "seed_inds": _c(labels["seed_inds"], torch.int32),
            "vote_label": _c(labels["vote_label"], f32),
            "vote_label_mask": _c(labels["vote_label_mask"], i64),
            "agg_xyz": _c(labels["aggregated_vote_xyz"], f32),
            "center_label": _c(labels["center_label"], f32),
            "box_label_mask": _c(labels["box_label_mask"], f32),
            "heading_class_label": _c(labels["heading_class_label"], i64),
            "heading_residual_label": _c(labels["heading_residual_label"], f32),
            "size_class_label": _c(labels["size_class_label"], i64),
            "size_residual_label": _c(labels["size_residual_label"], f32),
            "sem_cls_label": _c(labels["sem_cls_label"], i64),
            "mean_size_arr": array_const(np.asarray(cfg["mean_size_arr"], np.float32), dev),
        }
        B, K = t["center"].shape[:2]
        S = t["seed_xyz"].shape[1]
        G = t["center_label"].shape[1]
        VF = t["vote_xyz"].shape[1] // S
        t["objectness_label"] = torch.empty((B, K), dtype=i64, device=dev)
        t["objectness_mask"] = torch.empty((B, K), dtype=f32, device=dev)
        t["object_assignment"] = torch.empty((B, K), dtype=i64, device=dev)
        t["vote_arg"] = torch.empty((B, S), dtype=torch.int32, device=dev)
        t["center_g1"] = torch.empty((B, K), dtype=torch.int32, device=dev)
        t["center_k2"] = torch.empty((B, G), dtype=torch.int32, device=dev)
        lib = _C.load()
        lib.s2c_detection_loss_partial_floats.restype = _I
        t["partial"] = torch.empty((B, lib.s2c_detection_loss_partial_floats()), device=dev)
        t["stats"] = torch.empty(20, device=dev)
        w = cfg["objectness_cls_weights"]
        args = _Args(B, S, VF, t["vote_label"].shape[1], K, G, t["heading_scores"].shape[2],
                     t["size_scores"].shape[2], t["sem_cls_scores"].shape[2],
                     t["center_label"].shape[2], 0 if rows is None else rows.shape[2],
                     float(cfg["near_threshold"]),
                     float(cfg["far_threshold"]), float(w[0]), float(w[1]),
                     *[t[n].data_ptr() for n in _PTRS])
        with torch.cuda.device(dev):
            _C.call("s2c_detection_loss_fwd", ctypes.addressof(args), _C.stream_ptr())
        ctx.args, ctx.tensors, ctx.rows = args, t, rows
        outs = (t["stats"], t["objectness_label"], t["objectness_mask"],
                t["object_assignment"])
        ctx.mark_non_differentiable(*outs)
        ctx.set_materialize_grads(False)     # no zero fills for the non-differentiable outputs
        return (t["stats"][9].clone(),) + outs

    @staticmethod
    def backward(ctx, gdet, *_unused):
        if gdet is None:
            return (None,) * len(ctx.needs_input_grad)
        t, args, rows = ctx.tensors, ctx.args, ctx.rows
        dev = gdet.device
        gup = gdet.to(torch.float32).contiguous()
        if rows is not None:
            # one gradient tensor for the head rows: every column group is written by the
            # kernel (d centre lands in the centre-offset columns 2:5)
            d_rows = torch.empty_like(rows)
            off = {n: (t[n].data_ptr() - rows.data_ptr()) for n in _GRADS
                   if n not in ("vote_xyz", "center")}
            g_vote = torch.empty_like(t["vote_xyz"])
            ptr = {"vote_xyz": g_vote.data_ptr(), "center": d_rows.data_ptr() + 2 * 4}
            for n, o in off.items():
                ptr[n] = d_rows.data_ptr() + o
            grads = _Grads(*[ptr[n] for n in _GRADS])
            with torch.cuda.device(dev):
                _C.call("s2c_detection_loss_bwd", ctypes.addressof(args),
                        ctypes.addressof(grads), gup.data_ptr(), _C.stream_ptr())
            ctx.tensors = None
            # centre = aggregated_vote_xyz + rows[:, :, 2:5]: the same gradient for both
            d_agg = d_rows[:, :, 2:5] if ctx.needs_input_grad[11] else None
            return (g_vote,) + (None,) * 9 + (d_rows, d_agg)
        g = {n: torch.empty_like(t[n]) for n in _GRADS}
        grads = _Grads(*[g[n].data_ptr() for n in _GRADS])
        with torch.cuda.device(dev):
            _C.call("s2c_detection_loss_bwd", ctypes.addressof(args),
                    ctypes.addressof(grads), gup.data_ptr(), _C.stream_ptr())
        ctx.tensors = None
        return tuple(g[n] for n in _GRADS) + (None, None, None, None)


def available(data_dict):
    c = data_dict["center"]
    return (c.is_cuda and c.shape[1] <= 1024 and data_dict["center_label"].shape[1] <= 256
            and max(data_dict["heading_scores"].shape[2], data_dict["size_scores"].shape[2],
                    data_dict["sem_cls_scores"].shape[2]) <= 64)


def detection_loss(data_dict, config, near, far, cls_weights):
    """-> (det_total, stats, objectness_label, objectness_mask, object_assignment)."""
    d = data_dict
    cfg = {"mean_size_arr": config.mean_size_arr, "near_threshold": near,
           "far_threshold": far, "objectness_cls_weights": cls_weights}
    rows = d.get("_head_rows") if HEAD_ROWS_MODE else None
    if rows is not None and _views_of_rows(d, rows):
        det = lambda x: x.detach()
        return DetectionLoss.apply(
            d["vote_xyz"], det(d["objectness_scores"]), det(d["center"]),
            det(d["heading_scores"]), det(d["heading_residuals_normalized"]),
            det(d["size_scores"]), det(d["size_residuals_normalized"]),
            det(d["sem_cls_scores"]), d, cfg, rows, d["aggregated_vote_xyz"])
    return DetectionLoss.apply(
        d["vote_xyz"], d["objectness_scores"], d["center"], d["heading_scores"],
        d["heading_residuals_normalized"], d["size_scores"],
        d["size_residuals_normalized"], d["sem_cls_scores"], d, cfg)


HEAD_ROWS_MODE = True


def _views_of_rows(d, rows):
    """The score tensors are the column groups [0:2 | 2:5 centre | NH | NH | NS | NS*3 | NC]
    of the contiguous float32 head output `rows` (B,K,nout) (proposal_module.decode_scores)."""
    if not (rows.is_cuda and rows.dtype == torch.float32 and rows.is_contiguous()
            and rows.dim() == 3):
        return False
    B, K, nout = rows.shape
    NH, NS = d["heading_scores"].shape[2], d["size_scores"].shape[2]
    NC = d["sem_cls_scores"].shape[2]
    if nout != 5 + 2 * NH + 4 * NS + NC:
        return False
    cols = {"objectness_scores": 0, "heading_scores": 5,
            "heading_residuals_normalized": 5 + NH, "size_scores": 5 + 2 * NH,
            "size_residuals_normalized": 5 + 2 * NH + NS, "sem_cls_scores": 5 + 2 * NH + 4 * NS}
    base = rows.data_ptr()
    for n, c in cols.items():
        x = d[n]
        if x.dtype != torch.float32 or x.data_ptr() != base + 4 * c \
                or x.stride(0) != K * nout or x.stride(1) != nout:
            return False
    return True


# ---------------------------------------------------------------------------------------
# caption loss (loss_helper.py:189-230) -- csrc/s2c_loss.hip
_L64 = ctypes.c_longlong
_C.register("s2c_caption_loss_fwd", [_I, _I, _I, _P, _P, _L64, _P, _P, _P, _P, _P])
_C.register("s2c_caption_loss_bwd", [_I, _I, _I, _P, _P, _L64, _P, _P, _P, _P, _P, _P])


class CaptionLoss(Function):
    """(pred (B,T,V) f32, target (B,T) i64 view, good (B) bool) -> cap_loss, cap_acc."""

    @staticmethod
    def forward(ctx, pred, target, good):
        pred = pred if pred.is_contiguous() else pred.contiguous()
        B, T, V = pred.shape
        dev = pred.device
        if target.stride(1) != 1:
            target = target.contiguous()
        good8 = good.view(torch.uint8) if good.dtype == torch.bool else \
            (good != 0).view(torch.uint8)
        lse = torch.empty(B * T, dtype=torch.float32, device=dev)
        stats = torch.empty((B * T, 4), dtype=torch.float32, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _C.call("s2c_caption_loss_fwd", B, T, V, pred.data_ptr(), target.data_ptr(),
                    target.stride(0), good8.data_ptr(), lse.data_ptr(), stats.data_ptr(),
                    out.data_ptr(), _C.stream_ptr())
        ctx.save_for_backward(pred, target, good8, lse, out)
        loss, acc = out[0], out[1]
        ctx.mark_non_differentiable(acc)
        ctx.set_materialize_grads(False)
        return loss, acc

    @staticmethod
    def backward(ctx, g_loss, _g_acc):
        if g_loss is None:
            return None, None, None
        pred, target, good8, lse, out = ctx.saved_tensors
        B, T, V = pred.shape
        gup = g_loss.reshape(1).to(torch.float32).contiguous()
        dpred = torch.empty_like(pred)
        with torch.cuda.device(pred.device):
            _C.call("s2c_caption_loss_bwd", B, T, V, pred.data_ptr(), target.data_ptr(),
                    target.stride(0), good8.data_ptr(), lse.data_ptr(), out.data_ptr(),
                    gup.data_ptr(), dpred.data_ptr(), _C.stream_ptr())
        return dpred, None, None


def caption_loss_available(pred, target, good):
    return (pred.is_cuda and pred.dtype == torch.float32 and pred.dim() == 3
            and target.dtype == torch.int64 and good.dim() == 1)
