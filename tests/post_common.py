"""Shared by tests/gen_golden_post.py (reference side) and tests/test_post*.py."""
from types import SimpleNamespace

import numpy as np

B, K, N, NS, NC = 2, 48, 6000, 18, 18

# benchmark/predict.py:161-169 / lib/eval_helper.py:179-187 and two variations
POST_DICTS = {
    "predict": dict(remove_empty_box=True, use_3d_nms=True, nms_iou=0.25,
                    use_old_type_nms=False, cls_nms=True, per_class_proposal=True,
                    conf_thresh=0.05),
    "nms3d_nocls": dict(remove_empty_box=True, use_3d_nms=True, nms_iou=0.25,
                        use_old_type_nms=True, cls_nms=False, per_class_proposal=False,
                        conf_thresh=0.05),
    "nms2d": dict(remove_empty_box=False, use_3d_nms=False, nms_iou=0.25,
                  use_old_type_nms=False, cls_nms=False, per_class_proposal=False,
                  conf_thresh=0.3),
}


def mean_size_arr():
    return np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(NS, 3))


def dataset_config():
    return SimpleNamespace(mean_size_arr=mean_size_arr(), num_class=NC, num_heading_bin=1,
                           num_size_cluster=NS, axis_aligned=True)


def make_inputs(seed=11):
    """Head outputs of a VoteNet-like detector around 10 object clusters, so that boxes
    overlap (NMS has work), some are empty, and classes collide."""
    g = np.random.Generator(np.random.PCG64(seed))
    f32 = np.float32
    objs = g.uniform(-2.5, 2.5, size=(B, 10, 3)).astype(f32)
    pts = np.concatenate([
        objs[:, g.integers(0, 10, size=N // 2)] + g.normal(0, 0.25, size=(B, N // 2, 3)),
        g.uniform(-3, 3, size=(B, N - N // 2, 3))], 1).astype(f32)
    pc = np.concatenate([pts, g.normal(size=(B, N, 2)).astype(f32)], -1)   # (B,N,3+2)
    which = g.integers(0, 12, size=(B, K))
    far = which >= 10                                  # proposals in empty space
    center = np.where(far[..., None], g.uniform(5, 6, size=(B, K, 3)),
                      np.take_along_axis(objs, np.minimum(which, 9)[..., None].repeat(3, -1), 1)
                      + g.normal(0, 0.08, size=(B, K, 3))).astype(f32)
    size_scores = g.normal(size=(B, K, NS)).astype(f32)
    sem_scores = (g.normal(size=(B, K, NC)) + 3.0 * np.eye(NC)[which % 3]).astype(f32)
    return {
        "point_clouds": pc,
        "center": center,
        "heading_scores": g.normal(size=(B, K, 1)).astype(f32),
        "heading_residuals": (g.normal(size=(B, K, 1)) * 0.1).astype(f32),
        "size_scores": size_scores,
        "size_residuals": (g.normal(size=(B, K, NS, 3)) * 0.05).astype(f32),
        "sem_cls_scores": sem_scores,
        "objectness_scores": (g.normal(size=(B, K, 2)) * 2).astype(f32),
    }


def flatten(batch_pred_map_cls):
    """list of lists of (cls, corners, score) -> arrays."""
    scene, cls, corners, score = [], [], [], []
    for i, lst in enumerate(batch_pred_map_cls):
        for (c, box, s) in lst:
            scene.append(i)
            cls.append(int(c))
            corners.append(np.asarray(box, np.float64))
            score.append(float(s))
    return {"scene": np.asarray(scene, np.int64), "cls": np.asarray(cls, np.int64),
            "corners": np.asarray(corners, np.float64).reshape(-1, 8, 3),
            "score": np.asarray(score, np.float64)}
