"""Generates tests/golden/parse_predictions.npz by running the REFERENCE's
lib/ap_helper.py::parse_predictions (scipy Delaunay hull test, numpy NMS; imported from
/root/reference through oracle/ref_harness.py) on seeded synthetic head outputs, for
the POST_DICT of benchmark/predict.py:161-169 and two other NMS flavours.  The .npz
holds inputs + expected outputs only.

    python tests/gen_golden_post.py
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from tests import post_common as pc  # noqa: E402


def main():
    ref_harness.install()
    ap = importlib.import_module("lib.ap_helper")
    dc_mod = importlib.import_module("data.scannet.model_util_scannet")
    DC = dc_mod.ScannetDatasetConfig()
    DC.mean_size_arr = pc.mean_size_arr()
    inputs = pc.make_inputs(seed=11)
    out = {"in/" + k: v for k, v in inputs.items()}
    for name, post in pc.POST_DICTS.items():
        ep = {k: torch.from_numpy(v) for k, v in inputs.items()}
        cfg = dict(post, dataset_config=DC)
        res = ap.parse_predictions(ep, cfg)
        flat = pc.flatten(res)
        out[name + "/pred_mask"] = np.asarray(ep["pred_mask"], np.float64)
        for k, v in flat.items():
            out[name + "/" + k] = v
        print(name, "kept", int(ep["pred_mask"].sum()), "of", ep["pred_mask"].size,
              "entries", len(flat["scene"]))
    path = os.path.join(HERE, "golden", "parse_predictions.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
