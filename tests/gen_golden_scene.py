"""Generates tests/golden/scene_items.npz by running the REFERENCE's
lib/dataset.py::ScannetReferenceDataset.__getitem__ (imported from /root/reference through
oracle/ref_harness.py) on the seeded synthetic scenes of tests/scene_common.py.  The
dataset object is created without its file-loading constructor and handed the scene
arrays directly; `np.random.seed` fixes the draws, which tests replay through
oracle/scene_builder.py::draw.  The .npz holds expected outputs (+ the reference's mean
size table, a data file) only.

    python tests/gen_golden_scene.py
"""
import importlib
import multiprocessing as mp
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from tests import scene_common as sc  # noqa: E402


def main():
    ref_harness.install()
    ds = importlib.import_module("lib.dataset")
    out = {"mean_size_arr": np.asarray(ds.DC.mean_size_arr, np.float64),
           "nyu40id2class": np.array(sorted(ds.DC.nyu40id2class.items()), np.int64),
           "nyu40ids": np.asarray(ds.DC.nyu40ids, np.int64)}
    for name, (sseed, nv, npts, mvw, opts, rseed, oid) in sc.CASES.items():
        scene = sc.make_scene(sseed, nv, mvw)
        item = ds.ScannetReferenceDataset.__new__(ds.ScannetReferenceDataset)
        sid = "scene%04d_00" % sseed
        item.scanrefer = [{"scene_id": sid, "object_id": str(oid), "object_name": "chair",
                           "ann_id": "0", "token": ["a", "chair"]}]
        item.lang = {sid: {str(oid): {"0": np.zeros((32, 300))}}}
        item.lang_ids = {sid: {str(oid): {"0": np.zeros(32)}}}
        item.scene_data = {sid: {k: v for k, v in scene.items() if k != "multiview"}}
        item.multiview_data = {mp.current_process().pid: {sid: scene.get("multiview")}}
        item.raw2label = {"chair": 2}
        item.unique_multiple_lookup = {sid: {str(oid): {"0": 0}}}
        item.num_points = npts
        item.scan2cad_rotation = None
        if name.endswith("_rot"):
            item.scan2cad_rotation = {sid: {str(k): v.tolist() for k, v in
                                            sc.make_rotations(sseed, scene).items()}}
        for k, v in opts.items():
            setattr(item, k, v)
        np.random.seed(rseed)
        res = item[0]
        for k in sc.ITEM_KEYS:
            out[name + "/" + k] = np.asarray(res[k])
        print(name, "boxes", int(res["num_bbox"]), "voting points",
              int(res["vote_label_mask"].sum()), "ref", int(res["ref_box_label"].sum()),
              "cloud", res["point_clouds"].shape)
    # the test split's dataset (lib/dataset.py:542-617): vertices only
    sseed, nv, npts, mvw, opts, rseed = sc.TEST_CASE
    scene = sc.make_scene(sseed, nv, mvw)
    tds = ds.ScannetReferenceTestDataset.__new__(ds.ScannetReferenceTestDataset)
    sid = "scene%04d_00" % sseed
    tds.scanrefer_all_scene = [sid]
    tds.scene_data = {sid: {"mesh_vertices": scene["mesh_vertices"]}}
    tds.multiview_data = {mp.current_process().pid: {sid: scene.get("multiview")}}
    tds.glove = {"sos": np.zeros(300)}
    tds.num_points = npts
    for k, v in opts.items():
        setattr(tds, k, v)
    np.random.seed(rseed)
    res = tds[0]
    out["test_split/point_clouds"] = np.asarray(res["point_clouds"])
    print("test_split cloud", res["point_clouds"].shape)
    path = os.path.join(HERE, "golden", "scene_items.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
