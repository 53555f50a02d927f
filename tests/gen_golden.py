"""Generates tests/golden/capnet_{cfg1,c132}.npz by running the REFERENCE's CapNet
(imported from /root/reference through oracle/ref_harness.py) on seeded
synthetic inputs with deterministic weights.  Runs only where the reference
tree exists; the committed .npz holds inputs + expected outputs only.

    python tests/gen_golden.py [cfg1|c132|locals_all|well ...]      (default: cfg1)
    python tests/gen_golden.py search well 100 140 [salt ...]       (the draw `well` is made on)

c132 = the cfg3 channel layout (3+132) and proposal count (256) at N=8192; its inputs
are regenerated from the seed by the tests (pinned by `in_crc`), not stored.
"""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from tests import golden_common as gc  # noqa: E402


N_TRIALS = 3


def sensitivity(ref, model, sd, inputs, spec, msa, forced_inds, base):
    """Per stored key: the largest deviation (same metric as the tests: max |diff| over
    max(1, max|want|)) of the REFERENCE's own train-mode outputs / losses / gradients over
    N_TRIALS evaluations with one-rounding noise after every layer (golden_common.ulp_noise)
    and N_TRIALS with the input features perturbed below float32 resolution
    (golden_common.perturb_features), vote sampling held fixed.  The parity tests bound each key by
    max(1e-4, SENS_FACTOR x this) -- a measured conditioning bound instead of one blanket
    tolerance."""
    def rel(g, w):
        w = np.asarray(w, np.float64)
        return float(np.abs(np.asarray(g, np.float64) - w).max() / max(1.0, np.abs(w).max()))
    sens = {}
    for trial in range(2 * N_TRIALS):
        model.load_state_dict(sd)
        model.train()
        model.zero_grad()
        tin = gc.to_torch(inputs)
        if trial < N_TRIALS:        # one rounding error of noise after every layer
            probe = gc.ulp_noise(model, 1000 + trial)
        else:                       # ... or on the input features (coherent per point)
            probe = contextlib.nullcontext()
            tin["point_clouds"] = gc.perturb_features(tin["point_clouds"], 2000 + trial)
        with probe, gc.forced_vote_sampling(model, forced_inds):
            dd = model(tin, use_tf=True, is_eval=False)
            dd = ref.loss_helper.get_scene_cap_loss(
                dd, torch.device("cpu"), gc.LossConfig(msa), None, **gc.loss_flags(spec))
            dd["loss"].backward()
        cur = {}
        for k, v in gc.extract(dd, spec["train_keys"]).items():
            cur["train/" + k] = v
        for k in gc.loss_keys(spec):
            cur["loss/" + k] = np.asarray(dd[k].detach().cpu().numpy(), np.float64)
        for k, v in gc.extract_grads(model, spec).items():
            cur["grad/" + k] = v
        for k, v in cur.items():
            if base[k].dtype.kind == "f":
                sens[k] = max(sens.get(k, 0.0), rel(v, base[k]))
    return sens


def truth64(ref, model, sd, inputs, spec, msa, forced_inds):
    """The same train step once more in float64 (weights, inputs, every intermediate), the discrete
    geometry decisions taken on the float32 coordinates as the reference takes them
    (oracle/torch_ext.py) and the vote sampling forced: `truth/loss/*`, `truth/grad/*`.  The parity
    tests bound the HIP path's distance to this truth by a multiple of the float32 reference's OWN
    distance to it -- a measurement of what float32 can deliver per key, instead of a model."""
    model.load_state_dict(sd)
    model.double()
    model.train()
    model.zero_grad()
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        tin = {k: (v.double() if v.is_floating_point() else v) for k, v in gc.to_torch(inputs).items()}
        with gc.forced_vote_sampling(model, forced_inds):
            dd = model(tin, use_tf=True, is_eval=False)
            dd = ref.loss_helper.get_scene_cap_loss(
                dd, torch.device("cpu"), gc.LossConfig(msa), None, **gc.loss_flags(spec))
            dd["loss"].backward()
        out = {}
        for k in gc.loss_keys(spec):
            out["truth/loss/" + k] = np.asarray(dd[k].detach().cpu().numpy(), np.float64)
        for k, v in gc.extract_grads(model, spec).items():
            out["truth/grad/" + k] = np.asarray(v, np.float64)
    finally:
        torch.set_default_dtype(old)
        model.float()
        model.zero_grad()
    return out


def bn_conditioning(model, tin):
    """min over every train-mode BatchNorm channel of (batch variance / batch mean square) in one
    forward of the reference: a channel whose variance is tiny beside its mean is where one rounding
    of the statistics is amplified (the round-5 review's criterion for a well-conditioned fixture)."""
    worst = {}

    def hook(name):
        def fn(mod, inp):
            x = inp[0].detach().double()
            dims = [d for d in range(x.dim()) if d != 1]
            r = x.var(dims, unbiased=False) / (x * x).mean(dims).clamp_min(1e-300)
            worst[name] = float(r.min())
        return fn
    hooks = [m.register_forward_pre_hook(hook(n)) for n, m in model.named_modules()
             if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d))]
    model.train()
    with torch.no_grad():
        model(tin, use_tf=True, is_eval=False)
    for h in hooks:
        h.remove()
    return worst


def _rel(a, b):
    b = np.asarray(b, np.float64)
    return float(np.abs(np.asarray(a, np.float64) - b).max() / max(1.0, np.abs(b).max()))


def search(name="well", seeds=range(100, 140), salts=(0,)):
    """Looks for an input / weight draw on which the float32 reference agrees with its own float64
    evaluation on EVERY loss term and parameter gradient (no discrete decision within float32 noise of
    a tie) and every BatchNorm channel is well conditioned; prints one line per draw."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = ref_harness.reference_modules()
    spec = dict(gc.CFGS[name])
    msa = gc.mean_size_arr()
    ref.DC.mean_size_arr = msa
    for salt in salts:
        for seed in seeds:
            cfg = dict(spec["cfg"], seed=int(seed), weight_salt=int(salt))
            spec["cfg"] = cfg
            vocabulary, embeddings = gc.vocab_and_embeddings(cfg["V"])
            model = ref.capnet.CapNet(vocabulary=vocabulary, embeddings=embeddings,
                                      mean_size_arr=msa, **spec["kw"])
            sd = model.state_dict()
            with torch.no_grad():
                gc.det_fill_(sd, cfg.get("weight_salt", 0), cfg.get("well"))
            model.load_state_dict(sd)
            sd = {k: v.clone() for k, v in sd.items()}
            inputs = gc.make_inputs(cfg)
            model.train()
            with torch.no_grad():
                dry = model(gc.to_torch(inputs), use_tf=True, is_eval=False)
            inputs["ref_box_corner_label"][0] = dry["bbox_corner"][0, 5].numpy()
            model.load_state_dict(sd)
            model.zero_grad()
            dd = model(gc.to_torch(inputs), use_tf=True, is_eval=False)
            dd = ref.loss_helper.get_scene_cap_loss(
                dd, torch.device("cpu"), gc.LossConfig(msa), None, **gc.loss_flags(spec))
            dd["loss"].backward()
            base = {"loss/" + k: np.asarray(dd[k].detach().cpu().numpy(), np.float64)
                    for k in gc.loss_keys(spec)}
            base.update({"grad/" + k: v for k, v in gc.extract_grads(model, spec).items()})
            forced = dd["aggregated_vote_inds"].detach().clone()
            adj_ok = True
            if "adjacent_mat" in dd:
                adj = dd["adjacent_mat"]
                adj_ok = (adj * (dd["bbox_mask"] == 0).unsqueeze(1).float()).sum().item() == 0 \
                    and torch.diagonal(adj, dim1=1, dim2=2).sum().item() == 0
            good = int(dd["good_bbox_masks"].sum())
            truth = truth64(ref, model, sd, inputs, spec, msa, forced)
            far = sorted(((_rel(base[k[6:]], v), k[6:]) for k, v in truth.items()), reverse=True)
            free = [e for e in far if gc.decision_free(e[1])]
            print("   decision-free keys: worst %.2e (%s); pooled-stack keys: worst %.2e"
                  % (free[0][0], free[0][1], far[0][0]))
            model.load_state_dict(sd)
            bn = bn_conditioning(model, gc.to_torch(inputs))
            bn_min = min(bn.items(), key=lambda kv: kv[1])
            print("seed %d salt %d: worst |ref32-truth| %.2e (%s), 2nd %.2e, good %d, adj_ok %s, "
                  "bn min var/ms %.2e (%s)" % (seed, salt, far[0][0], far[0][1], far[1][0], good,
                                               adj_ok, bn_min[1], bn_min[0]), flush=True)
            if free[0][0] <= 1e-4 and far[0][0] <= 1e-3 and adj_ok and good >= 1:
                # a candidate: do six more float32 evaluations (one-ulp probes) agree as well?
                for k, v in gc.extract(dd, spec["train_keys"]).items():
                    base["train/" + k] = v
                sens = sensitivity(ref, model, sd, inputs, spec, msa, forced, base)
                worst = max(((v, k) for k, v in sens.items() if k.startswith(("grad/", "loss/"))
                             and gc.decision_free(k)))
                print("   CANDIDATE seed %d salt %d: probes' worst %.2e (%s)"
                      % (seed, salt, worst[0], worst[1]), flush=True)


def main(name="cfg1"):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = ref_harness.reference_modules()
    spec = gc.CFGS[name]
    cfg = spec["cfg"]
    vocabulary, embeddings = gc.vocab_and_embeddings(cfg["V"])
    msa = gc.mean_size_arr()
    # the reference's box decode uses the module-level DC (proposal_module.py:17);
    # give it the same mean sizes the ctor receives
    ref.DC.mean_size_arr = msa
    model = ref.capnet.CapNet(vocabulary=vocabulary, embeddings=embeddings,
                              mean_size_arr=msa, **spec["kw"])
    sd = model.state_dict()
    with torch.no_grad():
        gc.det_fill_(sd, cfg.get("weight_salt", 0), cfg.get("well"))
    model.load_state_dict(sd)
    sd = {k: v.clone() for k, v in sd.items()}  # detached copy (no aliasing)
    inputs = gc.make_inputs(cfg)

    # caption target of scene 0 := the reference's own proposal 5 (IoU 1 => a
    # "good" box that receives the caption loss); scene 1 keeps the random box
    model.train()
    with torch.no_grad():
        dry = model(gc.to_torch(inputs), use_tf=True, is_eval=False)
    inputs["ref_box_corner_label"][0] = dry["bbox_corner"][0, 5].numpy()
    model.load_state_dict(sd)

    out = {}
    model.train()
    model.zero_grad()
    dd = model(gc.to_torch(inputs), use_tf=True, is_eval=False)
    dd = ref.loss_helper.get_scene_cap_loss(
        dd, torch.device("cpu"), gc.LossConfig(msa), None, **gc.loss_flags(spec))
    dd["loss"].backward()
    for k in gc.loss_keys(spec):
        out["loss/" + k] = np.asarray(dd[k].detach().cpu().numpy(), np.float64)
    for k, v in gc.extract_grads(model, spec).items():
        out["grad/" + k] = v
    print("loss terms:", {k: float(out["loss/" + k]) for k in ("loss", "cap_loss", "ori_loss", "dist_loss", "cap_acc")
                          if "loss/" + k in out}, "good boxes:", int(dd["good_bbox_masks"].sum()))
    for k, v in gc.extract(dd, spec["train_keys"]).items():
        out["train/" + k] = v
    # sanity: the local top-k never had to pick among 1e30 ties
    nvalid = dd["bbox_mask"].sum(1)
    print("valid boxes per scene:", nvalid.tolist())
    if "adjacent_mat" in dd:
        print("edges src/tar:", dd["num_edge_source"].tolist(), dd["num_edge_target"].tolist())
        adj = dd["adjacent_mat"]
        bad = (adj * (dd["bbox_mask"] == 0).unsqueeze(1).float()).sum().item()
        diag = torch.diagonal(adj, dim1=1, dim2=2).sum().item()
        print("adjacency picks that hit invalid objects: %d, self picks: %d "
              "(both must be 0: no 1e30 tie-breaks)" % (bad, diag))
        assert bad == 0 and diag == 0
    sens = sensitivity(ref, model, sd, inputs, spec, msa,
                       dd["aggregated_vote_inds"].detach().clone(), dict(out))
    for k, v in sens.items():
        out["sens/" + k] = np.asarray(v, np.float64)
    truth = truth64(ref, model, sd, inputs, spec, msa, dd["aggregated_vote_inds"].detach().clone())
    out.update(truth)

    def rel(a, b):
        b = np.asarray(b, np.float64)
        return float(np.abs(np.asarray(a, np.float64) - b).max() / max(1.0, np.abs(b).max()))
    far = sorted(((rel(out[k[6:]], v), k[6:]) for k, v in truth.items()), reverse=True)
    print("float32 reference vs its float64 run, worst keys:", [(round(e, 6), k) for e, k in far[:8]])
    print("conditioning (one-ulp noise per layer), worst keys:",
          sorted(((round(v, 6), k) for k, v in sens.items()), reverse=True)[:8])
    if spec.get("strict"):
        # the well-conditioned fixture's contract, asserted where it is made
        model.load_state_dict(sd)
        bn = bn_conditioning(model, gc.to_torch(inputs))
        print("BatchNorm channels, min batch variance / mean square:", min(bn.values()))
        free = [e for e in far if gc.decision_free(e[1])]
        print("decision-free keys: %d of %d, worst |ref32 - truth| %.2e" % (len(free), len(far), free[0][0]))
        assert free[0][0] <= 1e-4, free[:4]
        assert max(v for k, v in sens.items() if k.startswith(("grad/", "loss/"))
                   and gc.decision_free(k)) <= 1e-4
        assert min(bn.values()) >= 1e-2, sorted(bn.items(), key=lambda kv: kv[1])[:4]
        out["bn_min_var_over_ms"] = np.asarray(min(bn.values()), np.float64)
    model.load_state_dict(sd)  # reset BN running stats touched by the train pass
    model.eval()
    with torch.no_grad():
        dd = model(gc.to_torch(inputs), use_tf=False, is_eval=True)
    for k, v in gc.extract(dd, spec["eval_keys"]).items():
        out["eval/" + k] = v
    if spec["store_inputs"]:
        for k, v in inputs.items():
            out["in/" + k] = v
    else:
        # the one input that is not a pure function of the seed (set from the dry run)
        out["in/ref_box_corner_label"] = inputs["ref_box_corner_label"]
        out["in_crc"] = np.asarray(gc.inputs_crc(inputs), np.int64)
    path = os.path.join(HERE, "golden", spec["file"])
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    if sys.argv[1:2] == ["search"]:
        lo, hi = (int(v) for v in (sys.argv[3:5] or ["100", "140"]))
        search(sys.argv[2] if len(sys.argv) > 2 else "well", range(lo, hi),
               tuple(int(v) for v in sys.argv[5:]) or (0,))
    else:
        for cfg_name in (sys.argv[1:] or ["cfg1"]):
            main(cfg_name)
