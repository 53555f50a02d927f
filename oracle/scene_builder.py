"""CPU restatement of the reference's training-item assembly.

TEST INFRASTRUCTURE ONLY: imported by tests/ and tools/bench_scene_builder.py's CPU
baseline leg, never by the product (scan2cap_amd/scene_builder.py runs HIP kernels and
fails loudly without them).

Follows `ScannetReferenceDataset.__getitem__` (lib/dataset.py:320-540) for ONE item,
with the `np.random` draws taken out into `draw()` so that the device path can be fed
the same numbers:

    point sampling            lib/dataset.py:365-368, utils/pc_utils.py:32-40
    colour / normal / multiview / height channels      lib/dataset.py:338-363
    box table                 lib/dataset.py:371-393
    augmentation              lib/dataset.py:396-426, :268-283 (`_translate`),
                              data/scannet/model_util_scannet.py:47-79
    votes                     lib/dataset.py:434-443
    size classes / residuals  lib/dataset.py:445-448
    reference box + corners   lib/dataset.py:451-477, model_util_scannet.py:156-172,
                              utils/box_util.py:340-383
    semantic classes, ids     lib/dataset.py:479-485
    Scan2CAD rotations        lib/dataset.py:490-503

Pinned by tests/gen_golden_scene.py, which runs the reference's own `__getitem__` in
this container on synthetic scenes (tests/golden/scene_items.npz).
"""
import numpy as np

MAX_NUM_OBJ = 128                                  # lib/dataset.py:27
MEAN_COLOR_RGB = np.array([109.8, 97.2, 83.8])     # lib/dataset.py:28
# nyu40 ids that carry votes / boxes (model_util_scannet.py:88)
NYU40IDS = np.array([3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21,
                     23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40])
# nyu40 id -> one of the 18 ScanRefer classes (model_util_scannet.py:100-115 applied to
# data/scannet/meta_data/scannetv2-labels.combined.tsv; "others" = 17)
NYU40ID2CLASS = {3: 0, 4: 1, 5: 2, 6: 3, 7: 4, 8: 5, 9: 6, 10: 7, 11: 8, 12: 9, 14: 10,
                 16: 11, 24: 12, 28: 13, 33: 14, 34: 15, 36: 16}
for _i in NYU40IDS:
    NYU40ID2CLASS.setdefault(int(_i), 17)


def _rot(axis, t):
    """utils/pc_utils.py:282-296 (`rotx`, `roty`) and `rotz`."""
    c, s = np.cos(t), np.sin(t)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    if axis == "y":
        return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def draw(num_vertices, num_points, augment, rng=np.random):
    """The random numbers of one item, in the reference's call order: the point choice
    (pc_utils.py:36-37), two flip coins, three angles (dataset.py:398-424), three
    translation factors (dataset.py:273-275)."""
    replace = num_vertices < num_points
    d = {"choices": rng.choice(num_vertices, num_points, replace=replace)}
    if augment:
        d["flip_x"] = bool(rng.random() > 0.5)
        d["flip_y"] = bool(rng.random() > 0.5)
        for ax in "xyz":
            d["rot_" + ax] = _rot(ax, (rng.random() * np.pi / 18) - np.pi / 36)
        grid = np.arange(-0.5, 0.501, 0.001)
        d["shift"] = np.array([rng.choice(grid, size=1)[0] for _ in range(3)])
    return d


def _rotate_boxes(boxes, rot, axis):
    """Axis-aligned cover of rotated axis-aligned boxes, as the reference computes it
    (model_util_scannet.py:47-79): the two half extents that the axis does not name
    are rotated as the x/y components of a corner vector."""
    half = boxes[:, 3:6] / 2.0
    a, b = {"x": (1, 2), "y": (0, 2), "z": (0, 1)}[axis]
    ext_a = np.full(boxes.shape[0], -np.inf)
    ext_b = np.full(boxes.shape[0], -np.inf)
    for sa, sb in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
        v = np.zeros((boxes.shape[0], 3))
        v[:, 0], v[:, 1] = sa * half[:, a], sb * half[:, b]
        v = np.dot(v, rot.T)
        ext_a, ext_b = np.maximum(ext_a, v[:, 0]), np.maximum(ext_b, v[:, 1])
    out = np.empty_like(boxes)
    out[:, 0:3] = np.dot(boxes[:, 0:3], rot.T)
    out[:, 3:6] = boxes[:, 3:6]
    out[:, 3 + a], out[:, 3 + b] = 2.0 * ext_a, 2.0 * ext_b
    return out


_CORNER_SIGNS = np.array([[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1],
                          [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]], np.float64)


def _corners(size, center):
    """utils/box_util.py:340-383 at heading 0 (ScanNet boxes are axis aligned,
    model_util_scannet.py:130-140): the rotation is the identity."""
    return _CORNER_SIGNS * (np.asarray(size)[..., None, :] / 2) + np.asarray(center)[..., None, :]


def _channels(scene, use_color, use_height, use_normal, use_multiview):
    """lib/dataset.py:338-363 (= :575-598 of the test dataset): xyz [rgb] [normal]
    [multiview] [height] for every vertex."""
    verts = scene["mesh_vertices"]
    cols = [verts[:, 0:3]]
    if use_color:
        # one application of the normalisation (the reference writes it back into its
        # cached scene through a view, dataset.py:342-344, so later epochs see it
        # applied repeatedly; items are pinned on first access)
        cols.append(((verts[:, 3:6] - MEAN_COLOR_RGB) / 256.0).astype(np.float32))
    if use_normal:
        cols.append(verts[:, 6:9])
    if use_multiview:
        cols.append(scene["multiview"])
    cloud = np.concatenate(cols, 1)
    if use_height:
        floor = np.percentile(cloud[:, 2], 0.99)
        cloud = np.concatenate([cloud, (cloud[:, 2] - floor)[:, None]], 1)
    return cloud


def build_test_item(scene, draws, use_color=False, use_height=True, use_normal=False,
                    use_multiview=False):
    """`ScannetReferenceTestDataset.__getitem__` (lib/dataset.py:567-609): the sampled,
    un-augmented cloud."""
    cloud = _channels(scene, use_color, use_height, use_normal, use_multiview)
    return {"point_clouds": cloud[draws["choices"]].astype(np.float32)}


def build_item(scene, draws, object_id, num_points, mean_size_arr, use_color=False,
               use_height=True, use_normal=False, use_multiview=False, augment=False,
               rotations=None):
    """scene: dict(mesh_vertices (Nv,>=6[9]) f32, instance_labels (Nv), semantic_labels
    (Nv), instance_bboxes (nb,8) f64 [cx cy cz dx dy dz nyu40id object_id], multiview
    (Nv,Cm) f32 when use_multiview).  rotations: {object_id: 3x3} (Scan2CAD) or None.
    Returns the tensor-valued entries of the reference's item dict."""
    boxes_in = scene["instance_bboxes"]
    cloud = _channels(scene, use_color, use_height, use_normal, use_multiview)
    choices = draws["choices"]
    cloud = cloud[choices]
    ins = scene["instance_labels"][choices]
    sem = scene["semantic_labels"][choices]

    nb = min(boxes_in.shape[0], MAX_NUM_OBJ)
    if nb == 0 or boxes_in.shape[0] > MAX_NUM_OBJ:
        # dataset.py:463-477 leaves gt_box_corner_label undefined / mis-shaped
        raise ValueError("a scene needs 1..%d boxes" % MAX_NUM_OBJ)
    boxes = np.zeros((MAX_NUM_OBJ, 6))
    boxes[:nb] = boxes_in[:nb, 0:6]
    box_mask = np.zeros(MAX_NUM_OBJ)
    box_mask[:nb] = 1

    if augment:
        if draws["flip_x"]:
            cloud[:, 0] = -1 * cloud[:, 0]
            boxes[:, 0] = -1 * boxes[:, 0]
        if draws["flip_y"]:
            cloud[:, 1] = -1 * cloud[:, 1]
            boxes[:, 1] = -1 * boxes[:, 1]
        for ax in "xyz":
            rot = draws["rot_" + ax]
            cloud[:, 0:3] = np.dot(cloud[:, 0:3], rot.T)
            boxes = _rotate_boxes(boxes, rot, ax)
        shift = [np.float64(v) for v in draws["shift"]]
        xyz = cloud[:, :3]
        xyz += shift
        boxes[:, :3] += shift

    votes = np.zeros((num_points, 3))
    votes_mask = np.zeros(num_points)
    for inst in np.unique(ins):
        rows = np.where(ins == inst)[0]
        if sem[rows[0]] in NYU40IDS:
            p = cloud[rows, :3]
            votes[rows] = 0.5 * (p.min(0) + p.max(0)) - p
            votes_mask[rows] = 1.0

    cls = np.array([NYU40ID2CLASS[int(v)] for v in boxes_in[:nb, 6]], np.int64)
    size_cls = np.zeros(MAX_NUM_OBJ)
    size_cls[:nb] = cls
    size_res = np.zeros((MAX_NUM_OBJ, 3))
    size_res[:nb] = boxes[:nb, 3:6] - mean_size_arr[cls]

    ref_label = np.zeros(MAX_NUM_OBJ)
    ref_center, ref_size_cls, ref_size_res = np.zeros(3), 0, np.zeros(3)
    ref_corners = np.zeros((8, 3))
    for i in range(nb):
        if boxes_in[i, 7] == object_id:
            ref_label[i] = 1
            ref_center, ref_size_cls, ref_size_res = boxes[i, 0:3], size_cls[i], size_res[i]
            ref_corners = _corners(mean_size_arr[int(ref_size_cls)] + ref_size_res, ref_center)
    gt_corners = np.zeros((MAX_NUM_OBJ, 8, 3))
    gt_corners[:nb] = _corners(mean_size_arr[cls] + size_res[:nb], boxes[:nb, 0:3])
    gt_masks = np.zeros(MAX_NUM_OBJ)
    gt_masks[:nb] = 1
    obj_ids = np.zeros(MAX_NUM_OBJ)
    obj_ids[:nb] = boxes_in[:nb, 7]
    sem_cls = np.zeros(MAX_NUM_OBJ)
    sem_cls[:nb] = cls

    obj_rot = np.zeros((MAX_NUM_OBJ, 3, 3))
    obj_rot_mask = np.zeros(MAX_NUM_OBJ)
    if rotations:
        for i, oid in enumerate(boxes_in[:nb, 7].astype(int)):
            if int(oid) in rotations:
                obj_rot[i] = np.asarray(rotations[int(oid)])
                obj_rot_mask[i] = 1

    zeros = np.zeros(MAX_NUM_OBJ)
    return {
        "point_clouds": cloud.astype(np.float32),
        "center_label": boxes.astype(np.float32)[:, 0:3],
        "heading_class_label": zeros.astype(np.int64),
        "heading_residual_label": zeros.astype(np.float32),
        "size_class_label": size_cls.astype(np.int64),
        "size_residual_label": size_res.astype(np.float32),
        "num_bbox": np.array(nb).astype(np.int64),
        "sem_cls_label": sem_cls.astype(np.int64),
        "scene_object_ids": obj_ids.astype(np.int64),
        "scene_object_rotations": obj_rot.astype(np.float32),
        "scene_object_rotation_masks": obj_rot_mask.astype(np.int64),
        "box_label_mask": box_mask.astype(np.float32),
        "vote_label": np.tile(votes, (1, 3)).astype(np.float32),
        "vote_label_mask": votes_mask.astype(np.int64),
        "ref_box_label": ref_label.astype(np.int64),
        "ref_center_label": np.asarray(ref_center).astype(np.float32),
        "ref_heading_class_label": np.array(0).astype(np.int64),
        "ref_heading_residual_label": np.array(0).astype(np.int64),
        "ref_size_class_label": np.array(int(ref_size_cls)).astype(np.int64),
        "ref_size_residual_label": np.asarray(ref_size_res).astype(np.float32),
        "ref_box_corner_label": ref_corners.astype(np.float64),
        "gt_box_corner_label": gt_corners.astype(np.float64),
        "gt_box_masks": gt_masks.astype(np.int64),
        "gt_box_object_ids": obj_ids.astype(np.int64),
    }
