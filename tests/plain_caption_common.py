"""Shared by tests/gen_golden_plain_caption.py and tests/test_plain_caption.py."""
import numpy as np

B, K, V, T, EVAL_LEN = 2, 12, 40, 9, 6


def make_inputs(embeddings, vocabulary, seed=3):
    g = np.random.Generator(np.random.PCG64(seed))
    f32 = np.float32
    centers = g.uniform(-2, 2, size=(B, K, 3))
    sizes = g.uniform(0.3, 1.2, size=(B, K, 3))
    sign = np.array([[1, 1, 1], [1, -1, 1], [-1, -1, 1], [-1, 1, 1],
                     [1, 1, -1], [1, -1, -1], [-1, -1, -1], [-1, 1, -1]], np.float64)
    corners = centers[:, :, None, :] + 0.5 * sizes[:, :, None, :] * sign[None, None]
    ref_box = corners[np.arange(B), [3, 7]] + g.normal(0, 0.02, size=(B, 8, 3))
    ids = g.integers(4, V, size=(B, T + 1))
    words = [vocabulary["idx2word"][str(i)] for i in ids.reshape(-1)]
    lang_feat = np.stack([embeddings[w] for w in words]).reshape(B, T + 1, 300).astype(f32)
    return {
        "bbox_feature": (g.standard_normal((B, K, 128)) * 0.5).astype(f32),
        "bbox_corner": corners.astype(np.float64),
        "ref_box_corner_label": ref_box.astype(np.float64),
        "lang_feat": lang_feat,
        "lang_len": np.array([T, T - 2], np.int64),
    }
