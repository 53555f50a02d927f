"""Shared by tests/gen_golden_scene.py (reference side) and tests/test_scene_builder*.py:
seeded synthetic ScanNet-like scenes in the layout of the reference's preprocessed files
(data/scannet/load_scannet_data.py:97-152: `_aligned_vert.npy` (Nv,9) f32 = xyz rgb
normal, `_ins_label.npy` / `_sem_label.npy` (Nv) uint32, `_aligned_bbox.npy` (nb,8) f64 =
centre, size, nyu40 id, object id) and the item cases the golden holds."""
import numpy as np

from scan2cap_amd.synthetic import make_scene  # noqa: F401  (shared with bench.py / smoke())

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
