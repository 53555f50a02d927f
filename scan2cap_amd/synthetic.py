"""Seeded synthetic scenes: no ScanNet/ScanRefer data is needed.

`scene_xyz` is shared by the parity tests and bench.py so both see the same
point distributions, including the adversarial rows the reference's tie/skip
rules care about (exact duplicates, |p|^2 <= 1e-3 points, all-zero rows).
"""
import numpy as np


def scene_xyz(batch, n, seed=42, mode="volume", adversarial=True):
    """(batch, n, 3) float32 room-like point sets.

    mode "volume": uniform in [-3,3]x[-3,3]x[0,2.5] m (ball queries never exit
    early = worst case).  mode "surface": floor + 4 walls + faces of 20 random
    boxes (denser balls, early exits like real scans).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((batch, n, 3), np.float32)
    for b in range(batch):
        if mode == "volume":
            p = rng.uniform([-3, -3, 0], [3, 3, 2.5], size=(n, 3))
        else:
            p = _surface(rng, n)
        p = p.astype(np.float32)
        if adversarial and n >= 256:
            k = max(1, min(64, n // 64))
            src = rng.integers(0, n, k)
            dst = rng.integers(0, n, k)
            p[dst] = p[src]                      # exact duplicates
            near = rng.integers(1, n, 8)
            p[near] = rng.uniform(-0.015, 0.015, size=(8, 3)).astype(np.float32)
            p[rng.integers(1, n, 8)] = 0.0        # all-zero rows
        out[b] = p
    return out


def _surface(rng, n):
    parts = []
    n_floor = n // 3
    f = rng.uniform([-3, -3, 0], [3, 3, 0], size=(n_floor, 3))
    parts.append(f)
    n_wall = n // 3
    w = rng.uniform([-3, -3, 0], [3, 3, 2.5], size=(n_wall, 3))
    side = rng.integers(0, 4, n_wall)
    w[side == 0, 0] = -3
    w[side == 1, 0] = 3
    w[side == 2, 1] = -3
    w[side == 3, 1] = 3
    parts.append(w)
    n_box = n - n_floor - n_wall
    centers = rng.uniform([-2.5, -2.5, 0.3], [2.5, 2.5, 1.2], size=(20, 3))
    sizes = rng.uniform(0.3, 1.2, size=(20, 3))
    which = rng.integers(0, 20, n_box)
    q = rng.uniform(-0.5, 0.5, size=(n_box, 3))
    face = rng.integers(0, 3, n_box)
    sign = rng.choice([-0.5, 0.5], n_box)
    q[np.arange(n_box), face] = sign
    parts.append(centers[which] + q * sizes[which])
    return np.concatenate(parts, 0)


MAX_NUM_OBJ = 128  # lib/dataset.py (center_label etc. are padded to 128 objects)


def scene_labels(xyz, num_boxes=32, seed=42, num_class=18, mean_size_arr=None):
    """Detection / relation labels with the shapes and dtypes of the reference's
    data_dict (lib/dataset.py:503-540): random axis-aligned
    boxes, votes from box membership (a point inside up to GT_VOTE_FACTOR=3 boxes
    votes for each centre; the first vote is repeated otherwise), identity
    object rotations.  xyz: (B,N,3) float32 numpy."""
    B, N, _ = xyz.shape
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    if mean_size_arr is None:
        mean_size_arr = np.ones((num_class, 3))
    out = dict(
        center_label=np.zeros((B, MAX_NUM_OBJ, 3), np.float32),
        heading_class_label=np.zeros((B, MAX_NUM_OBJ), np.int64),
        heading_residual_label=np.zeros((B, MAX_NUM_OBJ), np.float32),
        size_class_label=np.zeros((B, MAX_NUM_OBJ), np.int64),
        size_residual_label=np.zeros((B, MAX_NUM_OBJ, 3), np.float32),
        sem_cls_label=np.zeros((B, MAX_NUM_OBJ), np.int64),
        box_label_mask=np.zeros((B, MAX_NUM_OBJ), np.float32),
        vote_label=np.zeros((B, N, 9), np.float32),
        vote_label_mask=np.zeros((B, N), np.int64),
        scene_object_rotations=np.tile(np.eye(3, dtype=np.float32),
                                       (B, MAX_NUM_OBJ, 1, 1)),
        scene_object_rotation_masks=np.ones((B, MAX_NUM_OBJ), np.int64),
        gt_box_corner_label=np.zeros((B, MAX_NUM_OBJ, 8, 3), np.float64),
    )
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1])
    sy = np.array([1, -1, -1, 1, 1, -1, -1, 1])
    sz = np.array([1, 1, 1, 1, -1, -1, -1, -1])
    sign = np.stack([sx, sy, sz], -1)
    for b in range(B):
        lo, hi = xyz[b].min(0), xyz[b].max(0)
        centers = rng.uniform(lo + 0.3, hi - 0.3, size=(num_boxes, 3))
        sizes = rng.uniform(0.4, 1.4, size=(num_boxes, 3))
        cls = rng.integers(0, num_class, num_boxes)
        out["center_label"][b, :num_boxes] = centers
        out["size_class_label"][b, :num_boxes] = cls
        out["size_residual_label"][b, :num_boxes] = sizes - mean_size_arr[cls]
        out["sem_cls_label"][b, :num_boxes] = cls
        out["box_label_mask"][b, :num_boxes] = 1
        out["gt_box_corner_label"][b, :num_boxes] = \
            centers[:, None, :] + 0.5 * sizes[:, None, :] * sign[None]
        nvotes = np.zeros(N, np.int64)
        for k in range(num_boxes):
            inside = np.all(np.abs(xyz[b] - centers[k]) <= sizes[k] / 2, axis=1)
            ids = np.nonzero(inside)[0]
            votes = (centers[k] - xyz[b, ids]).astype(np.float32)
            for i, v in zip(ids, votes):
                j = nvotes[i]
                if j == 0:
                    out["vote_label"][b, i] = np.tile(v, 3)
                elif j < 3:
                    out["vote_label"][b, i, 3 * j:3 * j + 3] = v
                nvotes[i] += 1
            out["vote_label_mask"][b, ids] = 1
    return out


# ---- synthetic ScanNet-like scenes (the layout of the reference's preprocessed files) ----
# instances: nyu40 ids incl. wall(1) / floor(2) / ceiling(22), which carry neither
# votes nor boxes (lib/dataset.py:29, batch_load_scannet_data.py:42)
_SEM_POOL = [5, 5, 7, 4, 3, 39, 1, 2, 14, 33, 22, 24, 8, 40, 6, 12]
_OBJ_CLASS_IDS = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 23, 24,
                  25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40}


def make_scene(seed, num_vertices, multiview_width=0, num_instances=14):
    """One seeded scene as data/scannet/load_scannet_data.py:97-152 stores it:
    `mesh_vertices` (Nv,9) f32 = xyz rgb normal, `instance_labels` / `semantic_labels` (Nv)
    uint32, `instance_bboxes` (nb,8) f64 = centre, size, nyu40 id, object id [, `multiview`
    (Nv,W) f32].  Shared by bench.py --feed builder, smoke() and the scene-builder tests."""
    g = np.random.Generator(np.random.PCG64(seed))
    f32 = np.float32
    sem_of = np.array([_SEM_POOL[i % len(_SEM_POOL)] for i in range(num_instances)])
    centre = g.uniform([-3, -3, 0.2], [3, 3, 2.2], size=(num_instances, 3))
    extent = g.uniform(0.15, 0.6, size=(num_instances, 3))
    owner = g.integers(-3, num_instances, size=num_vertices)          # <0: unannotated
    xyz = g.uniform([-3.5, -3.5, 0.0], [3.5, 3.5, 2.6], size=(num_vertices, 3))
    has = owner >= 0
    xyz[has] = centre[owner[has]] + g.uniform(-1, 1, size=(has.sum(), 3)) * extent[owner[has]]
    rgb = g.integers(0, 256, size=(num_vertices, 3)).astype(np.float64)
    nrm = g.normal(size=(num_vertices, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    verts = np.concatenate([xyz, rgb, nrm], 1).astype(f32)
    ins = np.where(has, owner + 1, 0).astype(np.uint32)
    sem = np.where(has, sem_of[np.maximum(owner, 0)], 0).astype(np.uint32)
    boxes = []
    for i in range(num_instances):
        rows = np.where(ins == i + 1)[0]
        if len(rows) == 0 or int(sem_of[i]) not in _OBJ_CLASS_IDS:
            continue
        p = verts[rows, :3].astype(np.float64)
        lo, hi = p.min(0), p.max(0)
        boxes.append(np.concatenate([(lo + hi) / 2, hi - lo, [sem_of[i], i]]))
    scene = {"mesh_vertices": verts, "instance_labels": ins, "semantic_labels": sem,
             "instance_bboxes": np.asarray(boxes, np.float64)}
    if multiview_width:
        scene["multiview"] = np.maximum(
            g.normal(size=(num_vertices, multiview_width)) * 0.5, 0).astype(f32)
    return scene


def aim_reference_boxes_at_proposals(model, dd, proposal=0):
    """At random init no proposal overlaps the synthetic ground-truth box of a scene by
    IoU >= 0.25, so `good_bbox_masks` is all False, the caption loss is exactly 0
    (lib/loss_helper.py:189-230 masks it) and every gradient of the captioner and the relation
    graph is exactly 0.  Point each scene's described box (`ref_box_corner_label`) at the box
    the model itself predicts for one proposal: IoU = 1, the caption loss and its backward
    through decoder, attention and EdgeConv are live.  Returns the updated batch dict (the
    model's weights / BN statistics are left untouched)."""
    import torch
    state = {k: v.clone() for k, v in model.state_dict().items()}
    was_training = model.training
    with torch.no_grad():
        out = model(dict(dd), use_tf=True, is_eval=False)
    model.load_state_dict(state)
    model.train(was_training)
    dd = dict(dd)
    dd["ref_box_corner_label"] = out["bbox_corner"][:, proposal].detach().to(
        dd["ref_box_corner_label"].dtype).clone()
    return dd
