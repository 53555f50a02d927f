"""VoteNet + caption + relation losses: restatement of lib/loss_helper.py
(`get_scene_cap_loss` :381-491 and its helpers) and utils/nn_distance.py:13-59.

Same terms, weights and reductions as the reference, but every step is a
fixed-shape batched tensor op: no per-scene Python loops (:266-291, :330-347),
no boolean-mask indexing (dynamic shapes => host syncs, :216-224, :310-311) and
no `(B,N,M,3)` `.repeat` temporaries (nn_distance.py:47-49 -> broadcasting).
This is what seeds backward for the fwd+bwd metric.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .config import CONF
from . import loss_fused
from .consts import array_const, const

# CUDA tensors: detection terms from csrc/s2c_loss.hip (2 + 1 launches)
FUSED_DETECTION_LOSS = True
FUSED_CAPTION_LOSS = True        # masked CE + word accuracy: 3 launches (csrc/s2c_loss.hip)

FAR_THRESHOLD = 0.6
NEAR_THRESHOLD = 0.3
GT_VOTE_FACTOR = 3  # number of GT votes per point
OBJECTNESS_CLS_WEIGHTS = [0.2, 0.8]  # larger weight on positive objectness


def huber_loss(error, delta=1.0):
    """nn_distance.py:13-30."""
    abs_error = torch.abs(error)
    quadratic = torch.clamp(abs_error, max=delta)
    linear = abs_error - quadratic
    return 0.5 * quadratic ** 2 + delta * linear


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    """nn_distance.py:32-59.  pc1 (B,N,C), pc2 (B,M,C) ->
    dist1 (B,N), idx1 (B,N), dist2 (B,M), idx2 (B,M)."""
    pc_diff = pc1.unsqueeze(2) - pc2.unsqueeze(1)  # (B,N,M,C) by broadcasting
    if l1smooth:
        pc_dist = torch.sum(huber_loss(pc_diff, delta), dim=-1)
    elif l1:
        pc_dist = torch.sum(torch.abs(pc_diff), dim=-1)
    else:
        pc_dist = torch.sum(pc_diff ** 2, dim=-1)
    dist1, idx1 = torch.min(pc_dist, dim=2)
    dist2, idx2 = torch.min(pc_dist, dim=1)
    return dist1, idx1, dist2, idx2


def compute_vote_loss(data_dict):
    """loss_helper.py:24-69."""
    B, S = data_dict["seed_xyz"].shape[:2]
    vote_xyz = data_dict["vote_xyz"]
    seed_inds = data_dict["seed_inds"].long()
    seed_gt_votes_mask = torch.gather(data_dict["vote_label_mask"], 1, seed_inds)
    inds = seed_inds.view(B, S, 1).expand(B, S, 3 * GT_VOTE_FACTOR)
    seed_gt_votes = torch.gather(data_dict["vote_label"], 1, inds)
    seed_gt_votes = seed_gt_votes + data_dict["seed_xyz"].repeat(1, 1, 3)
    vote_r = vote_xyz.view(B * S, -1, 3)
    gt_r = seed_gt_votes.view(B * S, GT_VOTE_FACTOR, 3)
    _, _, dist2, _ = nn_distance(vote_r, gt_r, l1=True)
    votes_dist = torch.min(dist2, dim=1)[0].view(B, S)
    m = seed_gt_votes_mask.float()
    return torch.sum(votes_dist * m) / (torch.sum(m) + 1e-6)


def compute_objectness_loss(data_dict):
    """loss_helper.py:71-111."""
    aggregated_vote_xyz = data_dict["aggregated_vote_xyz"]
    gt_center = data_dict["center_label"][:, :, 0:3]
    dist1, ind1, _, _ = nn_distance(aggregated_vote_xyz, gt_center)
    euclidean_dist1 = torch.sqrt(dist1 + 1e-6)
    near = euclidean_dist1 < NEAR_THRESHOLD
    objectness_label = near.long()
    objectness_mask = (near | (euclidean_dist1 > FAR_THRESHOLD)).float()
    scores = data_dict["objectness_scores"]
    w = const("objectness_cls_weights", OBJECTNESS_CLS_WEIGHTS, scores.device)
    loss = F.cross_entropy(scores.transpose(2, 1), objectness_label, weight=w,
                           reduction="none")
    loss = torch.sum(loss * objectness_mask) / (torch.sum(objectness_mask) + 1e-6)
    return loss, objectness_label, objectness_mask, ind1


def compute_box_and_sem_cls_loss(data_dict, config):
    """loss_helper.py:113-187."""
    num_heading_bin = config.num_heading_bin
    num_size_cluster = config.num_size_cluster
    mean_size_arr = config.mean_size_arr
    object_assignment = data_dict["object_assignment"]
    dev = object_assignment.device

    pred_center = data_dict["center"]
    gt_center = data_dict["center_label"][:, :, 0:3]
    dist1, _, dist2, _ = nn_distance(pred_center, gt_center)
    box_label_mask = data_dict["box_label_mask"]
    objectness_label = data_dict["objectness_label"].float()
    denom = torch.sum(objectness_label) + 1e-6
    center_loss = torch.sum(dist1 * objectness_label) / denom + \
        torch.sum(dist2 * box_label_mask) / (torch.sum(box_label_mask) + 1e-6)

    heading_class_label = torch.gather(data_dict["heading_class_label"], 1, object_assignment)
    heading_class_loss = F.cross_entropy(
        data_dict["heading_scores"].transpose(2, 1), heading_class_label, reduction="none")
    heading_class_loss = torch.sum(heading_class_loss * objectness_label) / denom

    heading_residual_label = torch.gather(data_dict["heading_residual_label"], 1, object_assignment)
    heading_residual_normalized_label = heading_residual_label / (np.pi / num_heading_bin)
    heading_one_hot = F.one_hot(heading_class_label, num_heading_bin).float()
    heading_reg = huber_loss(
        torch.sum(data_dict["heading_residuals_normalized"] * heading_one_hot, -1)
        - heading_residual_normalized_label, delta=1.0)
    heading_reg_loss = torch.sum(heading_reg * objectness_label) / denom

    size_class_label = torch.gather(data_dict["size_class_label"], 1, object_assignment)
    size_class_loss = F.cross_entropy(
        data_dict["size_scores"].transpose(2, 1), size_class_label, reduction="none")
    size_class_loss = torch.sum(size_class_loss * objectness_label) / denom

    size_residual_label = torch.gather(
        data_dict["size_residual_label"], 1,
        object_assignment.unsqueeze(-1).expand(-1, -1, 3))
    size_one_hot = F.one_hot(size_class_label, num_size_cluster).float().unsqueeze(-1)
    predicted = torch.sum(data_dict["size_residuals_normalized"] * size_one_hot, 2)
    msa = array_const(np.asarray(mean_size_arr).astype(np.float32), dev)
    mean_size_label = torch.sum(size_one_hot * msa.view(1, 1, num_size_cluster, 3), 2)
    size_residual_label_normalized = size_residual_label / mean_size_label
    size_reg = torch.mean(huber_loss(predicted - size_residual_label_normalized,
                                     delta=1.0), -1)
    size_reg_loss = torch.sum(size_reg * objectness_label) / denom

    sem_cls_label = torch.gather(data_dict["sem_cls_label"], 1, object_assignment)
    sem_cls_loss = F.cross_entropy(
        data_dict["sem_cls_scores"].transpose(2, 1), sem_cls_label, reduction="none")
    sem_cls_loss = torch.sum(sem_cls_loss * objectness_label) / denom
    return (center_loss, heading_class_loss, heading_reg_loss, size_class_loss,
            size_reg_loss, sem_cls_loss)


def compute_cap_loss(data_dict, config, weights):
    """loss_helper.py:189-230 (cap_acc restated with masked sums)."""
    pred_caps = data_dict["lang_cap"]               # (B, T-1, V)
    num_words = data_dict.get("_num_words")
    if num_words is None:
        num_words = int(data_dict["lang_len"].max())
    target_caps = data_dict["lang_ids"][:, 1:num_words]
    V = pred_caps.shape[-1]
    if tuple(pred_caps.shape[:2]) != tuple(target_caps.shape):
        # a stale `_num_words` (static data_dict reused with new lang_len) or a
        # degenerate caption (num_words == 1: the decoder still runs one step)
        raise ValueError("caption loss: lang_cap has %s steps but the targets have %s "
                         "(is data_dict['_num_words'] = %r stale?)"
                         % (tuple(pred_caps.shape[:2]), tuple(target_caps.shape), num_words))
    if FUSED_CAPTION_LOSS and loss_fused.caption_loss_available(
            pred_caps, target_caps, data_dict["good_bbox_masks"]):
        return loss_fused.CaptionLoss.apply(pred_caps, target_caps,
                                            data_dict["good_bbox_masks"])
    ce = F.cross_entropy(pred_caps.reshape(-1, V), target_caps.reshape(-1),
                         ignore_index=0, reduction="none")
    good = data_dict["good_bbox_masks"]
    good_rep = good.unsqueeze(1).expand(-1, num_words - 1).reshape(-1)
    cap_loss = torch.sum(ce * good_rep) / (torch.sum(good_rep) + 1e-6)
    hit = (pred_caps.argmax(-1) == target_caps)
    count = (target_caps != 0) & good.unsqueeze(1)
    n = count.sum().float()
    cap_acc = torch.where(n > 0, (hit & count).sum().float() / n.clamp(min=1),
                          torch.zeros_like(n))
    return cap_loss, cap_acc


def radian_to_label(radians, num_bins=6):
    """loss_helper.py:232-247."""
    boundaries = torch.arange(np.pi / num_bins, np.pi - 1e-8, np.pi / num_bins,
                              device=radians.device)
    return torch.bucketize(radians, boundaries)


def _edge_slots(data_dict):
    """Positions [0, num_src*num_tar) of every scene's edge list, as a mask over
    the padded (B, K*L) axis (the reference slices per scene, :270-276)."""
    edge_indices = data_dict["edge_index"]
    M = data_dict["num_edge_source"] * data_dict["num_edge_target"]
    P = edge_indices.shape[-1]
    pos = torch.arange(P, device=edge_indices.device).view(1, P)
    live = pos < M.view(-1, 1)
    src = edge_indices[:, 0].long()
    tgt = edge_indices[:, 1].long()
    return live, src, tgt


def compute_node_orientation_loss(data_dict, num_bins=6):
    """loss_helper.py:250-313, all scenes at once."""
    object_assignment = data_dict["object_assignment"]
    edge_preds = data_dict["edge_orientations"]                   # (B,P,bins)
    B, K = object_assignment.shape
    rot = torch.gather(data_dict["scene_object_rotations"], 1,
                       object_assignment.view(B, K, 1, 1).expand(B, K, 3, 3))
    rot_masks = torch.gather(data_dict["scene_object_rotation_masks"], 1,
                             object_assignment)
    live, src, tgt = _edge_slots(data_dict)
    P = src.shape[1]
    source_rot = torch.gather(rot, 1, src.view(B, P, 1, 1).expand(B, P, 3, 3))
    target_rot = torch.gather(rot, 1, tgt.view(B, P, 1, 1).expand(B, P, 3, 3))
    relative = torch.matmul(source_rot, target_rot.transpose(3, 2))
    trace = torch.diagonal(relative, dim1=-2, dim2=-1).sum(-1)
    relative = torch.acos(torch.clamp(0.5 * (trace - 1), -1, 1))
    masks = torch.gather(rot_masks, 1, src) * torch.gather(rot_masks, 1, tgt)
    masks = masks * live.to(masks.dtype)
    labels = radian_to_label(relative, num_bins)
    ce = F.cross_entropy(edge_preds.reshape(B * P, -1), labels.reshape(-1),
                         reduction="none").view(B, P)
    loss = (ce * masks).sum() / (masks.sum() + 1e-8)
    hit = (edge_preds.argmax(-1) == labels) & (masks == 1)
    acc = hit.sum().float() / (masks.sum().float() + 1e-8)
    return loss, acc


def compute_node_distance_loss(data_dict):
    """loss_helper.py:315-353 (MSE over the concatenated live edges)."""
    gt_center = data_dict["center_label"][:, :, 0:3]
    object_assignment = data_dict["object_assignment"]
    gt_center = torch.gather(gt_center, 1,
                             object_assignment.unsqueeze(-1).expand(-1, -1, 3))
    live, src, tgt = _edge_slots(data_dict)
    B, P = src.shape
    sc = torch.gather(gt_center, 1, src.view(B, P, 1).expand(B, P, 3))
    tc = torch.gather(gt_center, 1, tgt.view(B, P, 1).expand(B, P, 3))
    labels = torch.norm(sc - tc, dim=2)
    err = (data_dict["edge_distances"] - labels) ** 2
    w = live.float()
    return (err * w).sum() / w.sum()


def _get_scene_cap_loss_fused(data_dict, device, config, weights, detection, caption,
                              orientation, distance, num_bins):
    """Same contract as below; detection terms from the fused HIP kernels."""
    det, stats, objectness_label, objectness_mask, object_assignment = \
        loss_fused.detection_loss(data_dict, config, NEAR_THRESHOLD, FAR_THRESHOLD,
                                  OBJECTNESS_CLS_WEIGHTS)
    data_dict["objectness_label"] = objectness_label
    data_dict["objectness_mask"] = objectness_mask
    data_dict["object_assignment"] = object_assignment
    data_dict["pos_ratio"], data_dict["neg_ratio"] = stats[10], stats[11]
    data_dict["obj_acc"] = stats[12]
    zero = torch.zeros((), device=device)
    for i, n in enumerate(loss_fused.STAT_NAMES):
        data_dict[n] = stats[i] if detection else zero
    if caption:
        data_dict["cap_loss"], data_dict["cap_acc"] = compute_cap_loss(
            data_dict, config, weights)
    else:
        data_dict["cap_loss"] = zero
        data_dict["cap_acc"] = zero
        data_dict["pred_ious"] = zero
    if orientation:
        data_dict["ori_loss"], data_dict["ori_acc"] = compute_node_orientation_loss(
            data_dict, num_bins)
    else:
        data_dict["ori_loss"] = zero
        data_dict["ori_acc"] = zero
    data_dict["dist_loss"] = compute_node_distance_loss(data_dict) if distance else zero
    if detection:
        loss = det + data_dict["cap_loss"] if caption else det
    else:
        loss = data_dict["cap_loss"]
    if orientation:
        loss = loss + 0.1 * data_dict["ori_loss"]
    if distance:
        loss = loss + 0.1 * data_dict["dist_loss"]
    data_dict["loss"] = loss
    _publish_parts(data_dict, det if detection else None, caption, orientation, distance)
    return data_dict


def _publish_parts(data_dict, det, caption, orientation, distance):
    """loss = _loss_det + _loss_rest (the detector part / everything that flows through the
    relation graph and the captioner) for parallel.backward_in_two_stages."""
    rest = None
    if caption:
        rest = data_dict["cap_loss"]
    if orientation:
        rest = 0.1 * data_dict["ori_loss"] if rest is None else rest + 0.1 * data_dict["ori_loss"]
    if distance:
        rest = 0.1 * data_dict["dist_loss"] if rest is None else rest + 0.1 * data_dict["dist_loss"]
    data_dict["_loss_det"], data_dict["_loss_rest"] = det, rest


def get_scene_cap_loss(data_dict, device, config, weights, detection=True,
                       caption=True, orientation=False, distance=False,
                       num_bins=CONF.TRAIN.NUM_BINS):
    """loss_helper.py:381-491: same keys written into data_dict, same weights."""
    if FUSED_DETECTION_LOSS and loss_fused.available(data_dict):
        return _get_scene_cap_loss_fused(data_dict, device, config, weights, detection,
                                         caption, orientation, distance, num_bins)
    vote_loss = compute_vote_loss(data_dict)
    objectness_loss, objectness_label, objectness_mask, object_assignment = \
        compute_objectness_loss(data_dict)
    total_num_proposal = objectness_label.shape[0] * objectness_label.shape[1]
    data_dict["objectness_label"] = objectness_label
    data_dict["objectness_mask"] = objectness_mask
    data_dict["object_assignment"] = object_assignment
    data_dict["pos_ratio"] = torch.sum(objectness_label.float()) / float(total_num_proposal)
    data_dict["neg_ratio"] = torch.sum(objectness_mask.float()) / float(total_num_proposal) \
        - data_dict["pos_ratio"]

    (center_loss, heading_cls_loss, heading_reg_loss, size_cls_loss,
     size_reg_loss, sem_cls_loss) = compute_box_and_sem_cls_loss(data_dict, config)
    box_loss = center_loss + 0.1 * heading_cls_loss + heading_reg_loss + \
        0.1 * size_cls_loss + size_reg_loss

    obj_pred_val = torch.argmax(data_dict["objectness_scores"], 2)
    data_dict["obj_acc"] = torch.sum(
        (obj_pred_val == objectness_label.long()).float() * objectness_mask) / \
        (torch.sum(objectness_mask) + 1e-6)

    zero = torch.zeros((), device=device)
    names = ("vote_loss", "objectness_loss", "center_loss", "heading_cls_loss",
             "heading_reg_loss", "size_cls_loss", "size_reg_loss", "sem_cls_loss",
             "box_loss")
    values = (vote_loss, objectness_loss, center_loss, heading_cls_loss,
              heading_reg_loss, size_cls_loss, size_reg_loss, sem_cls_loss, box_loss)
    for n, v in zip(names, values):
        data_dict[n] = v if detection else zero

    if caption:
        data_dict["cap_loss"], data_dict["cap_acc"] = compute_cap_loss(
            data_dict, config, weights)
    else:
        data_dict["cap_loss"] = zero
        data_dict["cap_acc"] = zero
        data_dict["pred_ious"] = zero
    if orientation:
        data_dict["ori_loss"], data_dict["ori_acc"] = compute_node_orientation_loss(
            data_dict, num_bins)
    else:
        data_dict["ori_loss"] = zero
        data_dict["ori_acc"] = zero
    data_dict["dist_loss"] = compute_node_distance_loss(data_dict) if distance else zero

    det = None
    if detection:
        loss = data_dict["vote_loss"] + 0.5 * data_dict["objectness_loss"] + \
            data_dict["box_loss"] + 0.1 * data_dict["sem_cls_loss"]
        loss = loss * 10  # amplify
        det = loss
        if caption:
            loss = loss + data_dict["cap_loss"]
    else:
        loss = data_dict["cap_loss"]
    if orientation:
        loss = loss + 0.1 * data_dict["ori_loss"]
    if distance:
        loss = loss + 0.1 * data_dict["dist_loss"]
    data_dict["loss"] = loss
    _publish_parts(data_dict, det, caption, orientation, distance)
    return data_dict
