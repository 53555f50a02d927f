"""CapNet forward against golden vectors produced by the REFERENCE's own
Python modules (tests/gen_golden.py; reference imported in the build container,
native ops supplied by the oracle).

* CPU flavour (not gpu): the build's host logic -- module tree, device-side box
  decode, batched graph / caption re-designs -- with the oracle injected as the
  op layer (a test double; the product never does this).
* GPU flavour (-m gpu): the real path, HIP kernels through the C ABI.

Tolerance 1e-4 of each tensor's scale on float outputs (north_star) unless a key is
listed in TOL_KEY with its measured bound; integer outputs exact.  Fixtures: cfg1
(BASELINE configs[0] shapes) and c132 (cfg3's 3+132 channels and 256 proposals at
N=8192).
"""
import os

import numpy as np
import pytest
import torch

from tests import golden_common as gc

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")

# Bound per key: |got - want| <= tol * max(1, max|want|).  Integer / bool outputs are always
# exact.  tol = 1e-4 (north_star) wherever that holds -- every eval-mode key, every
# train-mode key up to and including the votes -- and otherwise a MEASURED conditioning
# bound stored with the fixture: `sens/<key>` = how far the REFERENCE's own value of that
# key moves when each layer's output and gradient is perturbed by one float32 rounding
# error (tests/gen_golden.py: sensitivity; golden_common.ulp_noise, 3 trials), times
# golden_common.SENS_FACTOR (a re-ordered K-term dot product is off by up to ~sqrt(K)/2
# roundings).  Train-mode
# BatchNorm divides by per-batch standard deviations (13 layers; channels with tiny
# variance under the deterministic random weights) and its backward differences large
# sums, so one-ulp noise grows to ~1e-4 of scale behind the vote aggregation and to
# 0.3-2 % in the backbone's weight gradients -- on the reference's own arithmetic.  Any
# other correct fp32 evaluation order (the GPU kernels') lands within a small multiple.
TOL_DEFAULT = 1e-4
# Second bound for the train-mode losses and gradients (round 5): the fixture also holds the
# reference's train step evaluated in float64 (`truth/<key>`, tests/gen_golden.py: truth64 -- same
# weights, same discrete geometry decisions, same forced vote sampling).  The float32 reference is
# itself only an approximation of that truth, off by err_ref(key); the HIP path, another float32
# evaluation of the same mathematics, must land within K_TRUTH x max(err_ref(key), sens(key)) (or 1e-4)
# of the truth: the reference's ONE evaluation can sit closer to the truth than float32 guarantees
# (cfg1 `vgen.conv3.weight`: 1e-5 where the one-ulp probe moves it by 3e-4), and another summation
# order of the same kernels -- e.g. another persistent-grid size, which re-orders the BatchNorm
# partial sums -- lands 2e-4 away; so the measured conditioning is the floor of the bound, and the
# measured reference error widens it where the reference itself is far off (c132: 2.1e-2).
K_TRUTH = 3.0


def tol_of(sens, key):
    return max(TOL_DEFAULT, gc.SENS_FACTOR * sens.get(key, 0.0))


def build_model(device, name="cfg1"):
    from scan2cap_amd.models import CapNet
    spec = gc.CFGS[name]
    vocabulary, embeddings = gc.vocab_and_embeddings(spec["cfg"]["V"])
    model = CapNet(vocabulary=vocabulary, embeddings=embeddings,
                   mean_size_arr=gc.mean_size_arr(), **spec["kw"])
    sd = model.state_dict()
    with torch.no_grad():
        gc.det_fill_(sd, spec["cfg"].get("weight_salt", 0), spec["cfg"].get("well"))
    model.load_state_dict(sd)
    sd = {k: v.clone() for k, v in sd.items()}  # pristine copy for the eval pass
    return model.to(device), sd


class Report(object):
    """Collects every key's error before failing, so one run shows the whole table."""

    def __init__(self, device, name, sens=None, strict=False):
        self.device, self.name, self.rows, self.bad = device, name, {}, []
        self.sens = sens or {}
        # strict (the `well` fixture): NO conditioning floor on the decision-free keys
        # (golden_common.decision_free) -- 1e-4 against the float32 reference, and
        # max(1e-4, K_TRUTH x |ref32 - truth|) against its float64 run
        self.strict = strict

    def _floor(self, key):
        if self.strict and gc.decision_free(key):
            return 0.0
        return self.sens.get(key, 0.0)

    def check(self, got, want, key):
        g = got.detach().cpu().numpy()
        assert g.shape == want.shape, (key, g.shape, want.shape)
        if want.dtype.kind in "iub":
            ok = np.array_equal(g, want)
            self.rows[key] = {"exact": bool(ok)}
            if not ok:
                self.bad.append("%s: %d integer mismatches" % (key, int((g != want).sum())))
            return
        w = want.astype(np.float64)
        scale = max(1.0, float(np.abs(w).max()))
        err = float(np.abs(g.astype(np.float64) - w).max()) / scale
        tol = max(TOL_DEFAULT, gc.SENS_FACTOR * self._floor(key))
        self.rows[key] = {"rel_err": err, "tol": tol}
        if not err <= tol:
            self.bad.append("%s: %.3e of scale > %.1e" % (key, err, tol))

    def check_truth(self, got, ref32, truth, key):
        """|got - truth| <= max(1e-4, K_TRUTH * |ref32 - truth|) (both / max(1, max|truth|))."""
        t = np.asarray(truth, np.float64)
        scale = max(1.0, float(np.abs(t).max()))
        g = got.detach().cpu().numpy().astype(np.float64).reshape(t.shape)
        e_hip = float(np.abs(g - t).max()) / scale
        e_ref = float(np.abs(np.asarray(ref32, np.float64).reshape(t.shape) - t).max()) / scale
        bound = max(TOL_DEFAULT, K_TRUTH * max(e_ref, self._floor(key)))
        self.rows.setdefault(key, {}).update({"err_vs_truth": e_hip, "ref_vs_truth": e_ref,
                                              "truth_bound": bound})
        if not e_hip <= bound:
            self.bad.append("%s: %.3e from the float64 truth > %.1f x the reference's own %.3e"
                            % (key, e_hip, K_TRUTH, e_ref))

    def finish(self):
        out = os.environ.get("S2C_GOLDEN_REPORT")
        if out:
            import json
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "golden_report_%s_%s.json"
                                   % (self.name, self.device)), "w") as f:
                json.dump(self.rows, f, indent=1, sort_keys=True)
        assert not self.bad, "\n".join(self.bad)


def load_fixture(name):
    spec = gc.CFGS[name]
    ref = np.load(os.path.join(GOLDEN_DIR, spec["file"]))
    if spec["store_inputs"]:
        inputs = {k[3:]: ref[k] for k in ref.files if k.startswith("in/")}
    else:
        inputs = gc.make_inputs(spec["cfg"])
        inputs["ref_box_corner_label"] = ref["in/ref_box_corner_label"]
        assert gc.inputs_crc(inputs) == int(ref["in_crc"]), \
            "regenerated inputs differ from the ones the fixture was made with"
    return spec, ref, inputs


def check_vote_sampling(dd, rep, tag):
    """The vote-aggregation picks are exactly the oracle's FPS on the model's own votes."""
    from oracle import oracle as orc
    want = orc.furthest_point_sampling(
        np.ascontiguousarray(dd["vote_xyz"].detach().cpu().numpy()),
        dd["aggregated_vote_inds"].shape[1])
    ok = np.array_equal(dd["aggregated_vote_inds"].cpu().numpy(), want)
    rep.rows[tag + "/aggregated_vote_inds==oracle_fps(own vote_xyz)"] = {"exact": bool(ok)}
    if not ok:
        rep.bad.append(tag + ": vote FPS differs from the oracle on the same votes")


def check_local_masks(dd, want, rep, key, caption):
    """`valid_masks` is a DISCRETE function of the predicted boxes (the num_locals nearest objects
    of every proposal, an axis-aligned IoU threshold: caption_module.py:322-381): a last-bit
    difference in a box corner can legitimately move an object across the 10th / 11th-nearest
    boundary or the overlap threshold, after which the row differs by two entries from the
    fixture.  Like the vote sampling, it is therefore checked (a) exactly against the reference's
    own formulation evaluated on THIS run's boxes (torch ops, float64, CPU) and (b) against the
    fixture with at most 0.5 % of the rows differing."""
    from scan2cap_amd.models import graph_module as gm
    got = dd["valid_masks"].detach().cpu()
    corners = dd["bbox_corner"].detach().cpu()
    masks = dd["bbox_mask"].detach().cpu()
    B, K = masks.shape
    ids = torch.arange(K).view(1, K).expand(B, K)
    old = gm.USE_QUERY_KERNEL
    gm.USE_QUERY_KERNEL = False
    try:
        own, _ = gm.query_locals(corners, masks, ids, caption.num_locals, caption.query_mode,
                                 include_self=True)
    finally:
        gm.USE_QUERY_KERNEL = old
    exact = bool(torch.equal(got, own))
    rows_off = int((got.numpy() != want).any(-1).sum())
    rep.rows[key] = {"exact_on_own_boxes": exact, "rows_differing_from_fixture": rows_off,
                     "rows": int(B * K)}
    if not exact:
        rep.bad.append("%s: differs from the reference formulation on this run's own boxes" % key)
    if rows_off > max(1, (B * K) // 200):
        rep.bad.append("%s: %d of %d rows differ from the fixture" % (key, rows_off, B * K))


def run_and_compare(device, name="cfg1"):
    if os.environ.get("S2C_GOLDEN_OPBYOP") == "1":     # diagnosis: op-by-op GPU path
        from scan2cap_amd.opbyop import op_by_op
        with op_by_op():
            return _run_and_compare(device, name + "", tag="_opbyop")
    return _run_and_compare(device, name)


def _run_and_compare(device, name="cfg1", tag=""):
    spec, ref, inputs = load_fixture(name)
    model, sd = build_model(device, name)
    rep = Report(device, name + tag, {k[5:]: float(ref[k]) for k in ref.files
                                      if k.startswith("sens/")}, strict=bool(spec.get("strict")))

    from scan2cap_amd.loss_helper import get_scene_cap_loss
    model.train()
    with torch.no_grad():                      # free-running: sampling exactness
        check_vote_sampling(model(gc.to_torch(inputs, device), use_tf=True, is_eval=False),
                            rep, "train")
    model.load_state_dict(sd)
    model.zero_grad()
    with gc.forced_vote_sampling(model, torch.from_numpy(ref["train/aggregated_vote_inds"])):
        dd = model(gc.to_torch(inputs, device), use_tf=True, is_eval=False)
    for key, sl in spec["train_keys"].items():
        v = dd[key]
        rep.check(v[sl] if sl is not None else v, ref["train/" + key], "train/" + key)
    # loss restatement + backward (autograd through the custom grad kernels)
    dd = get_scene_cap_loss(dd, torch.device(device), gc.LossConfig(gc.mean_size_arr()),
                            None, **gc.loss_flags(spec))
    dd["loss"].backward()
    for key in gc.loss_keys(spec):
        rep.check(dd[key].detach().reshape(()), ref["loss/" + key].reshape(()),
                  "loss/" + key)
        if "truth/loss/" + key in ref.files:
            rep.check_truth(dd[key].detach().reshape(()), ref["loss/" + key],
                            ref["truth/loss/" + key], "loss/" + key)
    grads = gc.extract_grads(model, spec)
    if spec.get("grads") == "all":
        want = sorted(k[5:] for k in ref.files if k.startswith("grad/"))
        assert sorted(grads) == want, ("parameters with a gradient differ from the reference's",
                                       sorted(set(grads) ^ set(want)))
    for key, g in grads.items():
        rep.check(torch.from_numpy(g), ref["grad/" + key], "grad/" + key)
        if "truth/grad/" + key in ref.files:
            rep.check_truth(torch.from_numpy(g), ref["grad/" + key], ref["truth/grad/" + key],
                            "grad/" + key)

    model.load_state_dict(sd)
    model.eval()
    with torch.no_grad():
        free = model(gc.to_torch(inputs, device), use_tf=False, is_eval=True)
        check_vote_sampling(free, rep, "eval")
        if "eval/aggregated_vote_inds" in ref.files:
            with gc.forced_vote_sampling(
                    model, torch.from_numpy(ref["eval/aggregated_vote_inds"])):
                dd = model(gc.to_torch(inputs, device), use_tf=False, is_eval=True)
        else:
            dd = free
    for key, sl in spec["eval_keys"].items():
        v = dd[key]
        if key == "valid_masks" and sl is None and model.caption.num_locals != -1 and \
                not np.array_equal(v.detach().cpu().numpy(), ref["eval/" + key]):
            check_local_masks(dd, ref["eval/" + key], rep, "eval/" + key, model.caption)
            continue
        rep.check(v[sl] if sl is not None else v, ref["eval/" + key], "eval/" + key)
    # greedy tokens identical
    sl = spec["eval_keys"]["lang_cap"]
    lc = dd["lang_cap"][sl] if sl is not None else dd["lang_cap"]
    if not np.array_equal(lc.argmax(-1).cpu().numpy(), ref["eval/lang_cap"].argmax(-1)):
        rep.bad.append("eval/lang_cap: greedy tokens differ")
    rep.finish()


@pytest.mark.parametrize("name", sorted(gc.CFGS))
def test_capnet_host_logic_cpu(monkeypatch, name):
    from oracle import torch_ext
    from scan2cap_amd.pointnet2 import _ext
    for n in torch_ext.NAMES:
        monkeypatch.setattr(_ext, n, getattr(torch_ext, n))
    run_and_compare("cpu", name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(gc.CFGS))
def test_capnet_gpu(name):
    run_and_compare("cuda", name)
