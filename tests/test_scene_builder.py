"""the numpy restatement of the reference's item assembly
(oracle/scene_builder.py) against the outputs of the reference's own
`ScannetReferenceDataset.__getitem__` (tests/golden/scene_items.npz, generator:
tests/gen_golden_scene.py).  Everything is expected bit-exact: the restatement performs
the same numpy operations in the same order."""
import os

import numpy as np
import pytest

from oracle import scene_builder as osb
from tests import scene_common as sc

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "scene_items.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def oracle_item(name, golden):
    sseed, nv, npts, mvw, opts, rseed, oid = sc.CASES[name]
    scene = sc.make_scene(sseed, nv, mvw)
    np.random.seed(rseed)
    draws = osb.draw(nv, npts, opts["augment"])
    rot = sc.make_rotations(sseed, scene) if name.endswith("_rot") else None
    return osb.build_item(scene, draws, oid, npts, golden["mean_size_arr"],
                          rotations=rot, **opts)


@pytest.mark.parametrize("name", list(sc.CASES))
def test_oracle_matches_reference_item(name, golden):
    got = oracle_item(name, golden)
    for k in sc.ITEM_KEYS:
        want = golden[name + "/" + k]
        assert got[k].dtype == want.dtype, (k, got[k].dtype, want.dtype)
        assert got[k].shape == want.shape, (k, got[k].shape, want.shape)
        assert np.array_equal(got[k], want), (name, k, np.abs(
            got[k].astype(np.float64) - want.astype(np.float64)).max())


def test_oracle_matches_reference_test_split_item(golden):
    sseed, nv, npts, mvw, opts, rseed = sc.TEST_CASE
    scene = sc.make_scene(sseed, nv, mvw)
    np.random.seed(rseed)
    got = osb.build_test_item(scene, osb.draw(nv, npts, False), **opts)["point_clouds"]
    want = golden["test_split/point_clouds"]
    assert got.dtype == want.dtype and np.array_equal(got, want)


def test_nyu40_class_table_matches_reference(golden):
    # 37 voting ids, 18 classes, "others" = 17 (model_util_scannet.py:83-115)
    assert np.array_equal(golden["nyu40ids"], osb.NYU40IDS)
    assert sorted(osb.NYU40ID2CLASS.items()) == [tuple(r) for r in golden["nyu40id2class"]]


def test_scene_without_boxes_is_rejected(golden):
    scene = sc.make_scene(3, 500)
    scene["instance_bboxes"] = np.zeros((0, 8))
    np.random.seed(0)
    with pytest.raises(ValueError):
        osb.build_item(scene, osb.draw(500, 64, False), 0, 64, golden["mean_size_arr"])
