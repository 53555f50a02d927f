"""SURVEY §8 f3 on the GPU: items assembled by csrc/s2c_scene.hip (through the C ABI of
include/s2c_scene.h) against (1) the outputs of the reference's own `__getitem__`
(tests/golden/scene_items.npz), (2) the numpy oracle on batches over a multi-scene store,
(3) np.percentile for the floor height, and (4) size-independent properties at the cfg3
size (B=8, N=40000, 135 channels).

Tolerances: everything is compared bit-exactly (float32 and float64): the kernels perform
numpy's operations in numpy's order (float64 dgemm = fused multiply-add chain over k)."""
import os

import numpy as np
import pytest
import torch

from oracle import scene_builder as osb
from tests import scene_common as sc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "scene_items.npz")


def _sb():
    from scan2cap_amd import scene_builder
    return scene_builder


def _store(scenes, mvw, rotations=None):
    sb = _sb()
    store = sb.SceneStore("cuda:0", multiview_width=mvw)
    for i, s in enumerate(scenes):
        store.add_scene("s%d" % i, s["mesh_vertices"], s["instance_labels"],
                        s["semantic_labels"], s["instance_bboxes"], s.get("multiview"),
                        rotations[i] if rotations else None)
    return store.finalize()


def _assert_item(got, b, want, tag):
    for k in sc.ITEM_KEYS:
        g = got[k][b].cpu().numpy()
        w = np.asarray(want[k])
        assert g.dtype == w.dtype, (tag, k, g.dtype, w.dtype)
        assert g.shape == w.shape, (tag, k, g.shape, w.shape)
        if not np.array_equal(g, w):
            d = np.abs(g.astype(np.float64) - w.astype(np.float64))
            raise AssertionError("%s %s: %d of %d differ, max %g" %
                                 (tag, k, int((g != w).sum()), g.size, d.max()))


@pytest.mark.parametrize("name", list(sc.CASES))
def test_items_match_reference_golden(name):
    golden = np.load(GOLDEN)
    sseed, nv, npts, mvw, opts, rseed, oid = sc.CASES[name]
    scene = sc.make_scene(sseed, nv, mvw)
    rot = sc.make_rotations(sseed, scene) if name.endswith("_rot") else None
    store = _store([scene], mvw, [rot])
    builder = _sb().SceneBatchBuilder(store, golden["mean_size_arr"], num_points=npts, **opts)
    np.random.seed(rseed)
    draws = builder.draw(["s0"])
    out = builder.build(["s0"], [oid], draws)
    torch.cuda.synchronize()
    want = {k: golden[name + "/" + k] for k in sc.ITEM_KEYS}
    _assert_item(out, 0, want, name)


def test_test_split_item_matches_reference_golden():
    """`build_clouds` over a vertices-only store = the reference's test dataset item."""
    golden = np.load(GOLDEN)
    sseed, nv, npts, mvw, opts, rseed = sc.TEST_CASE
    scene = sc.make_scene(sseed, nv, mvw)
    sb = _sb()
    store = sb.SceneStore("cuda:0", multiview_width=mvw)
    store.add_scene("t", scene["mesh_vertices"], multiview=scene["multiview"])
    store.finalize()
    builder = sb.SceneBatchBuilder(store, np.ones((18, 3)), num_points=npts, **opts)
    np.random.seed(rseed)
    draws = builder.draw(["t"])
    got = builder.build_clouds(["t"], draws)["point_clouds"][0].cpu().numpy()
    assert np.array_equal(got, golden["test_split/point_clouds"])
    with pytest.raises(ValueError):
        builder.build(["t"], [0], draws)          # no labels in this store
    dev = builder.build_clouds(["t", "t"], builder.draw(["t", "t"], device_choices=True))
    ch = dev["_choices"]
    assert ch.unique(dim=1).shape[1] == npts or ch[0].unique().numel() == npts
    assert torch.equal(dev["point_clouds"][1, :, 3:6], store.verts[ch[1], 6:9])


@pytest.mark.parametrize("mvw,opts", [
    (16, dict(use_color=False, use_height=True, use_normal=True, use_multiview=True,
              augment=True)),
    (0, dict(use_color=True, use_height=True, use_normal=True, use_multiview=False,
             augment=True)),
    (0, dict(use_color=False, use_height=False, use_normal=False, use_multiview=False,
             augment=False)),
])
def test_batches_match_oracle(mvw, opts):
    golden = np.load(GOLDEN)
    msa = golden["mean_size_arr"]
    scenes = [sc.make_scene(20, 5000, mvw), sc.make_scene(21, 900, mvw),
              sc.make_scene(22, 12000, mvw, num_instances=40)]
    rots = [sc.make_rotations(20 + i, s) for i, s in enumerate(scenes)]
    store = _store(scenes, mvw, rots)
    N = 2048
    builder = _sb().SceneBatchBuilder(store, msa, num_points=N, **opts)
    ids = ["s2", "s0", "s1", "s2", "s0"]
    oids = [int(scenes[int(s[1])]["instance_bboxes"][k % 5, 7]) for k, s in enumerate(ids)]
    oids[2] = 12345            # no such object: reference labels stay zero
    rs = np.random.RandomState(77)
    draws = builder.draw(ids, rng=rs)
    out = builder.build(ids, oids, draws)
    # a second batch right behind the first (staging ring, no host sync in between)
    draws2 = builder.draw(ids[::-1], rng=rs)
    out2 = builder.build(ids[::-1], oids[::-1], draws2)
    torch.cuda.synchronize()
    for o, idl, oidl, dr in ((out, ids, oids, draws), (out2, ids[::-1], oids[::-1], draws2)):
        for b, sid in enumerate(idl):
            i = int(sid[1])
            want = osb.build_item(scenes[i], dr[b], oidl[b], N, msa, rotations=rots[i], **opts)
            _assert_item(o, b, want, "item %d (%s)" % (b, sid))
        assert o["heading_class_label"].dtype == torch.int64
        assert int(o["heading_class_label"].abs().sum()) == 0


def test_draw_consumes_numpy_state_like_the_oracle():
    scene = sc.make_scene(3, 3000)
    store = _store([scene], 0)
    builder = _sb().SceneBatchBuilder(store, np.ones((18, 3)), num_points=512, augment=True)
    np.random.seed(5)
    a = builder.draw(["s0", "s0"])
    tail_a = np.random.random()
    np.random.seed(5)
    b = [osb.draw(3000, 512, True), osb.draw(3000, 512, True)]
    tail_b = np.random.random()
    assert tail_a == tail_b
    for x, y in zip(a, b):
        assert sorted(x) == sorted(y)
        for k in x:
            assert np.array_equal(np.asarray(x[k]), np.asarray(y[k])), k


def test_device_side_vertex_sampling():
    """device_choices=True (s2c_scene_sample): N distinct in-range vertices per item (with
    replacement only for a scene smaller than N), every vertex about equally likely, no
    order bias, and the item built from them equals the oracle's for the same choices."""
    scenes = [sc.make_scene(30, 6000), sc.make_scene(31, 700), sc.make_scene(32, 4100)]
    store = _store(scenes, 0)
    N = 2048
    msa = np.ones((18, 3))
    builder = _sb().SceneBatchBuilder(store, msa, num_points=N, augment=True)
    ids = ["s0", "s1", "s2", "s0"]
    oids = [0, 0, 0, 1]
    rs = np.random.RandomState(1)
    hits = torch.zeros(6000, device="cuda")
    first = torch.zeros(6000, device="cuda")
    small_hits = torch.zeros(700, device="cuda")
    reps = 60
    for rep in range(reps):
        draws = builder.draw(ids, rng=rs, device_choices=True)
        out = builder.build(ids, oids, draws)
        ch = out["_choices"]
        assert ch.shape == (4, N) and ch.dtype == torch.int64
        for b, sid in enumerate(ids):
            nv = len(scenes[int(sid[1])]["mesh_vertices"])
            assert int(ch[b].min()) >= 0 and int(ch[b].max()) < nv
            if nv >= N:
                assert ch[b].unique().numel() == N
        assert not torch.equal(ch[0], ch[3])          # different seeds, same scene
        for b in (0, 3):
            hits += torch.bincount(ch[b], minlength=6000)
            first += torch.bincount(ch[b, :256], minlength=6000)
        small_hits += torch.bincount(ch[1], minlength=700)
    n = 2 * reps
    mean, sd = n * N / 6000.0, (n * (N / 6000.0) * (1 - N / 6000.0)) ** 0.5
    assert abs(float(hits.mean()) - mean) < 1e-3
    assert float(hits.min()) > mean - 6 * sd and float(hits.max()) < mean + 6 * sd
    assert 0.8 * sd < float(hits.std()) < 1.2 * sd            # binomial spread, not clumped
    m1 = n * 256 / 6000.0
    assert float(first.max()) < m1 + 6 * m1 ** 0.5 + 1        # no vertex favoured up front
    ms = reps * N / 700.0
    assert ms - 6 * ms ** 0.5 < float(small_hits.min()) and float(small_hits.max()) < ms + 6 * ms ** 0.5
    torch.cuda.synchronize()
    chn = out["_choices"].cpu().numpy()
    for b, sid in enumerate(ids):
        d = dict(draws[b], choices=chn[b])
        want = osb.build_item(scenes[int(sid[1])], d, oids[b], N, msa, augment=True)
        _assert_item(out, b, want, "device-sampled item %d" % b)


@pytest.mark.parametrize("nv,quant", [(1, 0), (2, 0), (3, 0), (101, 0), (1000, 8), (150001, 0),
                                      (262144, 64)])
def test_floor_height_matches_numpy_percentile(nv, quant):
    from scan2cap_amd import _C
    g = np.random.Generator(np.random.PCG64(nv))
    z = g.normal(0.3, 1.0, size=nv).astype(np.float32)
    if quant:
        z = (np.round(z * quant) / quant).astype(np.float32)      # many repeated values
    verts = np.zeros((nv, 9), np.float32)
    verts[:, 2] = z
    verts[:, 0] = g.normal(size=nv)
    d = torch.from_numpy(verts).cuda()
    out = torch.zeros(1, device="cuda")
    _C.call("s2c_scene_floor_height", nv, d.data_ptr(), 9, out.data_ptr(), _C.stream_ptr())
    want = np.percentile(verts[:, 2], 0.99)
    assert out.item() == float(want), (out.item(), float(want))


def test_cfg3_size_properties():
    """B=8 items of N=40000 points x 135 channels from ~150k-vertex scenes."""
    sb = _sb()
    mvw, N, B = 128, 40000, 8
    scenes = [sc.make_scene(40 + i, 150000 + 1111 * i, mvw, num_instances=60) for i in range(3)]
    store = _store(scenes, mvw)
    opts = dict(use_color=False, use_height=True, use_normal=True, use_multiview=True,
                augment=True)
    builder = sb.SceneBatchBuilder(store, np.full((18, 3), 0.5), num_points=N, **opts)
    ids = ["s%d" % (b % 3) for b in range(B)]
    oids = [int(scenes[b % 3]["instance_bboxes"][b, 7]) for b in range(B)]
    rs = np.random.RandomState(3)
    draws = builder.draw(ids, rng=rs)
    out = builder.build(ids, oids, draws)
    torch.cuda.synchronize()
    cloud = out["point_clouds"]
    assert cloud.shape == (B, N, 135)
    for b in range(B):
        s = b % 3
        rows = torch.from_numpy(draws[b]["choices"]).cuda() + int(store.vert_off_host[s])
        # copied channels are bit-identical to the resident scene
        assert torch.equal(cloud[b, :, 3:6], store.verts[rows, 6:9])
        assert torch.equal(cloud[b, :, 6:134], store.mv[rows])
        assert torch.equal(cloud[b, :, 134], store.verts[rows, 2] - store.floor[s])
        # augmentation is rigid: distances to the (transformed) origin are preserved
        raw = store.verts[rows, :3].double()
        shift = torch.from_numpy(draws[b]["shift"]).cuda()
        aug = cloud[b, :, :3].double() - shift
        assert torch.allclose(aug.norm(dim=1), raw.norm(dim=1), atol=2e-6, rtol=0)
        # votes: every voting point + its vote = the centre of its instance's sampled box
        ins = store.ins[rows].long()
        sem = store.sem[rows].long()
        vm = out["vote_label_mask"][b].bool()
        xyz = cloud[b, :, :3]
        lo = torch.full((sb.MAX_INSTANCE, 3), float("inf"), device="cuda")
        hi = -lo
        lo = lo.scatter_reduce(0, ins[:, None].expand(-1, 3), xyz, "amin")
        hi = hi.scatter_reduce(0, ins[:, None].expand(-1, 3), xyz, "amax")
        first = torch.full((sb.MAX_INSTANCE,), N, device="cuda").scatter_reduce(
            0, ins, torch.arange(N, device="cuda"), "amin")
        valid_id = torch.zeros(64, dtype=torch.bool, device="cuda")
        valid_id[list(sb.NYU40IDS)] = True
        want_mask = valid_id[sem[first.clamp(max=N - 1)]][ins]
        assert torch.equal(vm, want_mask)
        centre = (0.5 * (lo + hi))[ins]
        v = out["vote_label"][b]
        assert torch.equal(v[:, 0:3], v[:, 3:6]) and torch.equal(v[:, 0:3], v[:, 6:9])
        assert torch.equal(v[vm, 0:3], (centre - xyz)[vm])
        assert int(v[~vm].abs().sum()) == 0
    # boxes: labels of the described object agree with the per-box tables
    for b in range(B):
        r = int(out["ref_box_label"][b].argmax())
        assert int(out["scene_object_ids"][b, r]) == oids[b]
        assert torch.equal(out["ref_center_label"][b], out["center_label"][b, r])
        assert torch.equal(out["ref_box_corner_label"][b], out["gt_box_corner_label"][b, r])
        nb = int(out["num_bbox"][b])
        assert int(out["box_label_mask"][b].sum()) == nb == len(scenes[b % 3]["instance_bboxes"])


def test_builder_batch_trains_a_step_like_a_host_built_batch():
    """The device-built data_dict (+ AnnotationTable) drives the solver's step
    (lib/solver.py:293-302) exactly like the same items assembled on the host by the
    oracle and copied over (the reference's DataLoader route)."""
    import bench
    sb = _sb()
    B, N, K, V = 2, 4096, 64, 200
    wl = dict(B=B, N=N, C=4, K=K, V=V, train=True, desc="test")
    dev = torch.device("cuda")
    vocabulary, embeddings, table = bench.make_vocab(V)
    msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
    scenes = [sc.make_scene(60, 9000, num_instances=20), sc.make_scene(61, 7000,
                                                                       num_instances=20)]
    store = _store(scenes, 0)
    opts = dict(use_color=False, use_height=True, use_normal=True, use_multiview=False,
                augment=True)
    builder = sb.SceneBatchBuilder(store, msa, num_points=N, **opts)
    # annotations: 3 per scene
    g = np.random.Generator(np.random.PCG64(9))
    A, T = 6, 32
    lang_len = g.integers(8, T + 1, A).astype(np.int64)
    lang_ids = np.zeros((A, T), np.int64)
    for a in range(A):
        toks = [2] + list(g.integers(4, V, lang_len[a] - 2)) + [3]
        lang_ids[a, :len(toks)] = toks
    lang_feat = (table[lang_ids] * (lang_ids != 0)[..., None]).astype(np.float32)
    ann_scene = [0, 0, 0, 1, 1, 1]
    ann_obj = [int(scenes[s]["instance_bboxes"][a % 3, 7]) for a, s in enumerate(ann_scene)]
    anns = sb.AnnotationTable(dev, lang_feat, lang_ids, lang_len, ann_obj, np.arange(A) % 3,
                              np.full(A, 2), np.zeros(A))
    picks = [4, 1]
    ids = ["s%d" % ann_scene[a] for a in picks]
    oids = [ann_obj[a] for a in picks]
    rs = np.random.RandomState(11)
    draws = builder.draw(ids, rng=rs)
    dd = builder.build(ids, oids, draws)
    dd.update(anns.gather(picks))
    dd["_num_words"] = int(lang_len[picks].max())
    assert torch.equal(dd["object_id"], torch.as_tensor(oids, device=dev))
    # host route
    items = [osb.build_item(scenes[ann_scene[a]], draws[b], ann_obj[a], N, msa, **opts)
             for b, a in enumerate(picks)]
    host = {k: torch.from_numpy(np.stack([it[k] for it in items])).to(dev) for k in items[0]}
    host.update(lang_feat=torch.from_numpy(lang_feat[picks]).to(dev),
                lang_ids=torch.from_numpy(lang_ids[picks]).to(dev),
                lang_len=torch.from_numpy(lang_len[picks]).to(dev),
                _num_words=dd["_num_words"])
    torch.manual_seed(0)
    model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True,
                           fused=True)
    step = bench.make_step(model, wl, bench.LossConfig(msa), opt, None, dev)
    got = float(step(dd).detach())
    model.load_state_dict(state)
    opt2 = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True,
                            fused=True)
    step2 = bench.make_step(model, wl, bench.LossConfig(msa), opt2, None, dev)
    want = float(step2(host).detach())
    assert np.isfinite(got)
    np.testing.assert_allclose(got, want, rtol=1e-5)


def test_batch_feeder_fills_static_buffer_sets():
    """BatchFeeder (the producer behind `bench.py --feed builder`): three static buffer sets,
    host-paced hand-over; every set ends up holding exactly the batch a direct `build` with
    the same draws gives, including after the sets have been recycled."""
    sb = _sb()
    scenes = [sc.make_scene(70 + i, 5000 + 500 * i, num_instances=20) for i in range(3)]
    store = _store(scenes, 0)
    N, B = 1024, 2
    msa = np.full((18, 3), 0.7)
    builder = sb.SceneBatchBuilder(store, msa, num_points=N, use_normal=True, augment=True)
    ids0 = ["s0", "s1"]
    oid = lambda ids: [int(scenes[int(s[1])]["instance_bboxes"][0, 7]) for s in ids]
    template = builder.build(ids0, oid(ids0), builder.draw(ids0, rng=np.random.RandomState(0)))
    keys = [k for k in template if not k.startswith("_")]
    sets = [{k: torch.zeros_like(template[k]) for k in keys} for _ in range(3)]
    feeder = sb.BatchFeeder(builder, None, sets, device_choices=True,
                            rng=np.random.RandomState(5))
    mirror = np.random.RandomState(5)
    picks = [["s%d" % ((k + b) % 3) for b in range(B)] for k in range(7)]
    want = []
    for k, ids in enumerate(picks):
        p = k % 3
        feeder.produce(p, ids, oid(ids), host_wait=True)
        feeder.acquire(p)                                  # consumer side
        snap = {key: sets[p][key].clone() for key in keys}
        feeder.release(p)
        draws = builder.draw(ids, rng=mirror, device_choices=True)
        ref = builder.build(ids, oid(ids), draws)
        want.append((snap, ref))
    torch.cuda.synchronize()
    for k, (snap, ref) in enumerate(want):
        for key in keys:
            assert torch.equal(snap[key], ref[key]), (k, key)
    with pytest.raises(ValueError):
        builder.build(ids0, oid(ids0), builder.draw(ids0),
                      out={"point_clouds": torch.zeros(1, device="cuda")})
