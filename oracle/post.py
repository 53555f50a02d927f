"""CPU restatement (numpy, float64) of `parse_predictions`, lib/ap_helper.py:40-178,
with utils/nms.py:38-151 and utils/box_util.py:340-358.  TEST INFRASTRUCTURE ONLY.

Differences in form, not in value: the per-box scipy Delaunay hull test
(model_util_scannet.py:13-22) is restated as interval tests in the box frame (the
boxes are cuboids rotated about Y); `tests/gen_golden_post.py` pins this file against
the reference's own function, Delaunay included.
"""
import numpy as np


def softmax(x):
    """ap_helper.py:33-38."""
    p = np.exp(x - np.max(x, axis=-1, keepdims=True))
    return p / np.sum(p, axis=-1, keepdims=True)


def roty(t):
    """box_util.py:325-331."""
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def get_3d_box(box_size, heading_angle, center):
    """box_util.py:340-358."""
    l, w, h = box_size
    x = [l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2]
    y = [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2]
    z = [h / 2, h / 2, h / 2, h / 2, -h / 2, -h / 2, -h / 2, -h / 2]
    c = np.dot(roty(heading_angle), np.vstack([x, y, z]))
    return (c + np.asarray(center, np.float64).reshape(3, 1)).T


def points_in_box(pc, center, size, angle):
    d = pc.astype(np.float64) - np.asarray(center, np.float64)
    ca, sa = np.cos(angle), np.sin(angle)
    lx = ca * d[:, 0] - sa * d[:, 2]
    lz = sa * d[:, 0] + ca * d[:, 2]
    return (np.abs(lx) <= size[0] / 2) & (np.abs(d[:, 1]) <= size[1] / 2) & \
        (np.abs(lz) <= size[2] / 2)


def nms(boxes, score, cls, thresh, old_type, add_eps):
    """Greedy NMS in descending score (nms.py:72-151); boxes (n,6)."""
    n = boxes.shape[0]
    area = np.prod(boxes[:, 3:6] - boxes[:, 0:3], axis=1)
    order = sorted(range(n), key=lambda j: (score[j], j))       # ascending, ties by index
    alive = np.ones(n, bool)
    pick = []
    for i in reversed(order):
        if not alive[i]:
            continue
        pick.append(i)
        alive[i] = False
        for j in range(n):
            if not alive[j]:
                continue
            ext = np.maximum(0.0, np.minimum(boxes[i, 3:6], boxes[j, 3:6]) -
                             np.maximum(boxes[i, 0:3], boxes[j, 0:3]))
            inter = ext[0] * ext[1] * ext[2]
            if old_type:
                o = inter / area[j]
            else:
                o = inter / (area[i] + area[j] - inter + (1e-8 if add_eps else 0.0))
            if cls is not None and cls[i] != cls[j]:
                o = 0.0
            if o > thresh:
                alive[j] = False
    return pick


def parse_predictions(end_points, config_dict, heading_angle_fn=None):
    """end_points: numpy arrays.  Returns (batch_pred_map_cls, pred_mask)."""
    cfg = config_dict["dataset_config"]
    center = end_points["center"]
    B, K = center.shape[:2]
    hcls = np.argmax(end_points["heading_scores"], -1)
    hres = np.take_along_axis(end_points["heading_residuals"], hcls[..., None], 2)[..., 0]
    scls = np.argmax(end_points["size_scores"], -1)
    sres = np.take_along_axis(end_points["size_residuals"],
                              scls[..., None, None].repeat(3, -1), 2)[:, :, 0]
    sem = np.argmax(end_points["sem_cls_scores"], -1)
    sem_probs = softmax(end_points["sem_cls_scores"])
    obj_prob = softmax(end_points["objectness_scores"])[:, :, 1]
    msa = np.asarray(cfg.mean_size_arr, np.float64)
    corners = np.zeros((B, K, 8, 3))
    size = np.zeros((B, K, 3))
    angle = np.zeros((B, K))
    for i in range(B):
        for j in range(K):
            angle[i, j] = heading_angle_fn(hcls[i, j], hres[i, j]) if heading_angle_fn else 0.0
            size[i, j] = msa[scls[i, j]] + sres[i, j]
            corners[i, j] = get_3d_box(size[i, j], angle[i, j], center[i, j])
    nonempty = np.ones((B, K), bool)
    if config_dict["remove_empty_box"]:
        pc = end_points["point_clouds"][:, :, 0:3]
        for i in range(B):
            for j in range(K):
                nonempty[i, j] = points_in_box(pc[i], center[i, j], size[i, j],
                                               angle[i, j]).sum() >= 5
    pred_mask = np.zeros((B, K))
    lo, hi = corners.min(2), corners.max(2)
    for i in range(B):
        ids = np.where(nonempty[i])[0]
        assert len(ids) > 0
        if not config_dict["use_3d_nms"]:
            bx = np.stack([lo[i, :, 0], lo[i, :, 2], np.zeros(K), hi[i, :, 0], hi[i, :, 2],
                           np.ones(K)], -1)
            cls, eps = None, False
        else:
            bx = np.concatenate([lo[i], hi[i]], -1)
            same = config_dict.get("cls_nms", False)
            cls, eps = (sem[i][ids] if same else None), same
        pick = nms(bx[ids], obj_prob[i][ids].astype(np.float64), cls,
                   config_dict["nms_iou"], config_dict["use_old_type_nms"], eps)
        pred_mask[i, ids[pick]] = 1
    out = []
    for i in range(B):
        sel = [j for j in range(K)
               if pred_mask[i, j] == 1 and obj_prob[i, j] > config_dict["conf_thresh"]]
        if config_dict["per_class_proposal"]:
            cur = []
            for ii in range(cfg.num_class):
                cur += [(ii, corners[i, j], sem_probs[i, j, ii] * obj_prob[i, j]) for j in sel]
        else:
            cur = [(int(sem[i, j]), corners[i, j], obj_prob[i, j]) for j in sel]
        out.append(cur)
    return out, pred_mask
