"""Checkpoint compatibility with the reference's files.

* build container only (skipped where /root/reference is absent, i.e. on the GPU box): the
  reference's real `pretrained/PRETRAIN_VOTENET_*/model.pth` load into this build's modules
  through the reference's own call (scripts/train.py:96-105) with NO missing detector key,
  and produce the same detections as the reference's modules holding the same weights;
* -m gpu: `checkpoint.tar` (lib/solver.py:501-510) saved from a hipGraph-captured training
  run and resumed (scripts/train.py:138-145) into (a) the live captured graph, in place,
  and (b) a fresh process-like model + optimizer with a re-captured graph -- both continue
  the original run.
"""
import glob
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tests import golden_common as gc  # noqa: E402

REF_PRETRAINED = "/root/reference/pretrained"
_PRETRAINED = sorted(glob.glob(os.path.join(REF_PRETRAINED, "PRETRAIN_VOTENET_*", "model.pth")))


def _flags_of(dirname):
    tail = dirname.replace("PRETRAIN_VOTENET_XYZ", "")
    return dict(use_color="_COLOR" in tail, use_multiview="_MULTIVIEW" in tail,
                use_normal="_NORMAL" in tail)


@pytest.mark.skipif(not _PRETRAINED, reason="reference pretrained weights not present")
@pytest.mark.parametrize("path", _PRETRAINED, ids=lambda p: os.path.basename(os.path.dirname(p)))
def test_pretrained_votenet_loads_with_the_reference_call(path):
    from scan2cap_amd import checkpoint as ck
    from scan2cap_amd.models import CapNet
    flags = _flags_of(os.path.basename(os.path.dirname(path)))
    assert ck.pretrained_name(**flags) == os.path.basename(os.path.dirname(path))
    vocabulary, embeddings = gc.vocab_and_embeddings(40)
    model = CapNet(18, vocabulary, embeddings, 1, 18, gc.mean_size_arr(),
                   input_feature_dim=ck.input_channels(**flags), num_proposal=256,
                   num_locals=10, use_topdown=True, graph_mode="edge_conv",
                   num_graph_steps=2, use_relation=True)
    before = model.caption.classifier.weight.detach().clone()
    missing, unexpected = ck.mount_pretrained_votenet(model, path, no_detection=True)
    assert [k for k in missing if k.startswith(ck.PRETRAINED_PREFIXES)] == [], missing
    assert list(unexpected) == [], unexpected
    sd = torch.load(path, map_location="cpu")
    got = model.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    assert all(not p.requires_grad for p in model.backbone_net.parameters())
    assert all(not p.requires_grad for p in model.proposal.parameters())
    assert model.caption.classifier.weight.requires_grad
    assert torch.equal(model.caption.classifier.weight, before)


@pytest.mark.skipif(not _PRETRAINED, reason="reference pretrained weights not present")
def test_pretrained_weights_give_the_reference_detections(monkeypatch):
    """Real weights, reference modules vs this build's modules, CPU (oracle ops as the
    op layer of both): BASELINE configs[0] (1 scene, XYZ+height, N=4096, 32 proposals)."""
    from oracle import ref_harness, torch_ext
    from scan2cap_amd.models import CapNet
    from scan2cap_amd.pointnet2 import _ext
    from scan2cap_amd.synthetic import scene_xyz
    if not ref_harness.available():
        pytest.skip("reference tree not present")
    path = os.path.join(REF_PRETRAINED, "PRETRAIN_VOTENET_XYZ", "model.pth")
    ref = ref_harness.reference_modules()
    msa = ref.DC.mean_size_arr
    kw = dict(num_class=18, vocabulary=None, embeddings=None, num_heading_bin=1,
              num_size_cluster=18, mean_size_arr=msa, input_feature_dim=1, num_proposal=32,
              no_caption=True)
    theirs = ref.capnet.CapNet(**kw).eval()
    ours = CapNet(**kw).eval()
    sd = torch.load(path, map_location="cpu")
    r1 = theirs.load_state_dict(sd, strict=False)
    r2 = ours.load_state_dict(sd, strict=False)
    assert list(r1.missing_keys) == list(r2.missing_keys) == []
    for n in torch_ext.NAMES:
        monkeypatch.setattr(_ext, n, getattr(torch_ext, n))
    xyz = scene_xyz(1, 4096, seed=11, mode="surface")
    height = xyz[..., 2:3] - np.percentile(xyz[..., 2], 0.99)
    pc = torch.from_numpy(np.concatenate([xyz, height], -1).astype(np.float32))
    with torch.no_grad():
        want = theirs({"point_clouds": pc.clone()})
        got = ours({"point_clouds": pc.clone()})
    for k in ("sa1_inds", "aggregated_vote_inds", "bbox_mask"):
        assert torch.equal(got[k].long(), want[k].long()), k
    for k in ("fp2_features", "vote_xyz", "vote_features", "objectness_scores", "center",
              "size_scores", "size_residuals", "sem_cls_scores", "bbox_feature"):
        err = float((got[k] - want[k]).abs().max() / max(1.0, float(want[k].abs().max())))
        assert err <= 1e-4, (k, err)
    np.testing.assert_allclose(got["bbox_corner"].numpy(),
                               np.asarray(want["bbox_corner"].detach().cpu()
                                          if torch.is_tensor(want["bbox_corner"])
                                          else want["bbox_corner"]), atol=1e-4)


# ---------------------------------------------------------------------------------------
def _train_setup(seed=0):
    import bench
    wl = dict(B=2, N=4096, C=4, K=64, V=200, train=True, desc="test")
    dev = torch.device("cuda")
    vocabulary, embeddings, table = bench.make_vocab(wl["V"])
    msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
    torch.manual_seed(seed)
    model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).train()
    dd = bench.to_device(bench.make_batch(wl, wl["B"], 7, table, msa), dev)
    return bench, wl, model, dd, bench.LossConfig(msa), dev


def _graphed(bench, wl, model, dd, cfg, dev):
    """model + fused capturable Adam + the captured step (geometry in static slots), as
    bench.py runs it."""
    from scan2cap_amd.graphs import GraphedCallable
    from scan2cap_amd.pipeline import GeometrySlots
    from scan2cap_amd.optim import FusedAdam          # (round 6: bench.py's optimizer; the reference
    opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)    # resume below loads its checkpoint
    step = bench.make_step(model, wl, cfg, opt, None, dev)             # into a torch.optim.Adam)
    slots = GeometrySlots(model.backbone_net, dd["point_clouds"], 1)

    def body():
        d = dict(dd)
        d["_geometry"] = slots.geometry(0)
        return step(d)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    g = GraphedCallable(body).capture()          # warms up with real steps ...
    model.load_state_dict(state)                 # ... so restore weights / fresh Adam state
    for st in opt.state.values():
        for v in st.values():
            if torch.is_tensor(v):
                v.zero_()
    slots.refill(0, dd["point_clouds"])

    def run(n):
        out = []
        for _ in range(n):
            slots.acquire(0)
            out.append(float(g().detach()))
            slots.release(0)
            slots.refill(0, dd["point_clouds"])
        torch.cuda.synchronize()
        return out
    return opt, run


LR = 1e-3


def _snapshot(model, opt):
    w = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = {id(p): n for n, p in model.named_parameters()}
    st = {names[id(p)]: {k: v.detach().clone() for k, v in s.items() if torch.is_tensor(v)}
          for p, s in opt.state.items() if s}
    return w, st


def _assert_continues(model, opt, loss, ref_loss, ref_w, ref_st, what):
    """One step after a resume vs the original run's same step.  The loss is a function of
    the restored weights / BN statistics only; the weights and Adam moments after the step
    additionally prove the optimizer state was restored.  (Trajectories are NOT compared
    beyond one step: at random init one last-bit difference of the float atomics moves a
    discrete label assignment, cf. tests/test_train_loop_gpu.py.)"""
    np.testing.assert_allclose(loss, ref_loss, rtol=1e-4, err_msg=what)
    w, st = _snapshot(model, opt)
    for k, b in ref_w.items():
        if not b.is_floating_point():
            assert torch.equal(w[k], b), (what, k)
            continue
        d = (w[k] - b).abs()
        # Adam moves a weight by ~lr per step whatever the gradient's size: an entry whose
        # gradient is rounding noise (e.g. a BN bias in front of another BN) may step the
        # other way (2 lr apart); most entries must agree closely
        assert float(d.max()) <= 2.5 * LR + 1e-4 * float(b.abs().max()), (what, k)
        assert float((d > 0.1 * LR).float().mean()) <= 0.5, (what, k)
    assert set(st) == set(ref_st), what
    for n, s in ref_st.items():
        assert float(st[n]["step"]) == float(s["step"]), (what, n)
        # (a bias in front of a BatchNorm has a zero gradient: its moments are rounding
        # noise of ~1e-7, hence the absolute floors)
        for k, floor in (("exp_avg", 1e-6), ("exp_avg_sq", 1e-10)):
            scale = float(s[k].abs().max())
            # (float atomics: two runs of the same step differ in the last bits, and at random
            # init a flipped arg-max moves single entries by ~1e-3 of the scale; moments that
            # were NOT restored would be off by the scale itself)
            assert float((st[n][k] - s[k]).abs().max()) <= 1e-2 * scale + floor, (what, n, k)


@pytest.mark.gpu
def test_checkpoint_tar_resumes_a_graphed_run(tmp_path):
    from scan2cap_amd import checkpoint as ck
    bench, wl, model, dd, cfg, dev = _train_setup()
    opt, run = _graphed(bench, wl, model, dd, cfg, dev)
    run(2)
    root = str(tmp_path / "outputs" / "stamp")
    ck.save_checkpoint(root, 0, model, opt, {"epoch": 0, "sum": -1.0})
    # the reference's file layout (lib/solver.py:501-510)
    saved = torch.load(os.path.join(root, "checkpoint.tar"), map_location="cpu")
    assert sorted(saved) == ["best", "epoch", "model_state_dict", "optimizer_state_dict"]
    assert os.path.exists(os.path.join(root, "model_last.pth"))
    (want,) = run(1)                                         # step 3 of the original run
    w_want, st_want = _snapshot(model, opt)

    # (a) resume INTO the live captured graph: weights and Adam state copied in place
    for p in model.parameters():
        p.data.add_(0.05)                                   # wreck the live state first
    for s in opt.state.values():
        for v in s.values():
            if torch.is_tensor(v):
                v.add_(1.0)
    epoch, best = ck.load_checkpoint(root, model, opt, inplace=True)
    assert epoch == 0 and best["sum"] == -1.0
    (got,) = run(1)
    _assert_continues(model, opt, got, want, w_want, st_want, "in-place resume, live graph")

    # (b) a fresh model + optimizer (different init) with its own captured graph, resumed
    # in place
    bench2, wl2, model2, dd2, cfg2, dev2 = _train_setup(seed=123)
    opt2, run2 = _graphed(bench2, wl2, model2, dd2, cfg2, dev2)
    ck.load_checkpoint(root, model2, opt2, inplace=True)
    (got,) = run2(1)
    _assert_continues(model2, opt2, got, want, w_want, st_want, "fresh process, in place")

    # (c) the reference's own resume calls (scripts/train.py:138-145: optimizer state
    # REPLACED), then the step is built / captured afterwards
    bench3, wl3, model3, dd3, cfg3, dev3 = _train_setup(seed=321)
    ckpt = torch.load(os.path.join(root, "checkpoint.tar"))
    model3.load_state_dict(ckpt["model_state_dict"])
    opt3 = torch.optim.Adam(model3.parameters(), lr=LR, weight_decay=1e-5, capturable=True,
                            fused=True)
    opt3.load_state_dict(ckpt["optimizer_state_dict"])
    step3 = bench3.make_step(model3, wl3, cfg3, opt3, None, dev3)
    got = float(step3(dd3).detach())
    _assert_continues(model3, opt3, got, want, w_want, st_want, "reference resume calls")
