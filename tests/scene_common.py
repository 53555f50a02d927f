"""Shared by tests/gen_golden_scene.py (reference side) and tests/test_scene_builder*.py:
seeded synthetic ScanNet-like scenes in the layout of the reference's preprocessed files
(data/scannet/load_scannet_data.py:97-152: `_aligned_vert.npy` (Nv,9) f32 = xyz rgb
normal, `_ins_label.npy` / `_sem_label.npy` (Nv) uint32, `_aligned_bbox.npy` (nb,8) f64 =
centre, size, nyu40 id, object id) and the item cases the golden holds."""
import numpy as np

# instances: nyu40 ids incl. wall(1) / floor(2) / ceiling(22), which carry neither
# votes nor boxes (lib/dataset.py:29, batch_load_scannet_data.py:42)
_SEM_POOL = [5, 5, 7, 4, 3, 39, 1, 2, 14, 33, 22, 24, 8, 40, 6, 12]
_OBJ_CLASS_IDS = {3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 23, 24,
                  25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40}

CASES = {
    # name: (scene seed, Nv, num_points, multiview width, options, np.random seed, object id)
    "plain": (3, 3000, 1024, 0,
              dict(use_color=False, use_height=True, use_normal=False, use_multiview=False,
                   augment=False), 100, 4),
    "aug_mv": (4, 3000, 1024, 8,
               dict(use_color=False, use_height=True, use_normal=True, use_multiview=True,
                    augment=True), 101, 2),
    "aug_color_small": (5, 700, 1024, 0,
                        dict(use_color=True, use_height=True, use_normal=False,
                             use_multiview=False, augment=True), 102, 0),
    "aug_mv128_rot": (6, 2500, 512, 128,
                      dict(use_color=False, use_height=True, use_normal=True,
                           use_multiview=True, augment=True), 103, 9),
    "noheight_missing_ref": (7, 2000, 768, 0,
                             dict(use_color=False, use_height=False, use_normal=True,
                                  use_multiview=False, augment=True), 104, 999),
}


# (scene seed, Nv, num_points, multiview width, options, np.random seed) of the TEST dataset
TEST_CASE = (8, 2600, 1024, 8, dict(use_color=False, use_height=True, use_normal=True,
                                    use_multiview=True), 105)


def make_scene(seed, num_vertices, multiview_width=0, num_instances=14):
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


def make_rotations(seed, scene):
    """Scan2CAD-style {object id: 3x3} for every second box (lib/dataset.py:495-503)."""
    g = np.random.Generator(np.random.PCG64(1000 + seed))
    out = {}
    for oid in scene["instance_bboxes"][::2, 7].astype(int):
        q, _ = np.linalg.qr(g.normal(size=(3, 3)))
        out[int(oid)] = q
    return out


ITEM_KEYS = ("point_clouds", "center_label", "heading_class_label", "heading_residual_label",
             "size_class_label", "size_residual_label", "num_bbox", "sem_cls_label",
             "scene_object_ids", "scene_object_rotations", "scene_object_rotation_masks",
             "box_label_mask", "vote_label", "vote_label_mask", "ref_box_label",
             "ref_center_label", "ref_heading_class_label", "ref_heading_residual_label",
             "ref_size_class_label", "ref_size_residual_label", "ref_box_corner_label",
             "gt_box_corner_label", "gt_box_masks", "gt_box_object_ids")
