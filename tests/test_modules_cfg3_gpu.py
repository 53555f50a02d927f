"""Every differentiable module of the detector in ISOLATION at its cfg3 shape
(BASELINE.json configs[2]: B=8, N=40000, C=132, K=256), forward AND backward, against a
float64 evaluation of the reference's formulation.

Why this file exists (round-2 review): the model-level gradient check of
tests/test_configs_gpu.py bounds each gradient by a multiple of the model's measured
conditioning, which for the backbone weights at cfg3 is 10-20 % of the tensor's scale --
a wrong term in a hand-written backward kernel would pass there.  Conditioning is a property
of the CHAIN (1e8 arg-maxes and ReLU masks in series); one module with its inputs, its
ball-query rows and its upstream gradient held fixed is well conditioned.  So: run the fused
cfg3 step once, record for every module what it was given and the gradient it received, then
evaluate the module three ways from those identical inputs --

  (a) the fused HIP path (what bench.py runs: gather-fused streaming GEMM at M = 1 048 576
      rows with the 540-byte row stride of C = 132, pooled-layer algebra, s2c_bn_bwd_gemm,
      scatter kernels ...),
  (b) the op-by-op fp32 formulation (QueryAndGroup -> Conv2d/BatchNorm2d/ReLU -> max_pool2d,
      pointnet2_modules.py:226-257 / :371-416, voting_module.py:33-60,
      proposal_module.py:46-54) over the nine `_ext` ops,
  (c) a float64 torch evaluation of (b),

with ONE refinement: a module holds ~1e6..1e8 ReLU thresholds and max-pool arg-maxes, and a
float64 evaluation decides a handful of them differently from ANY float32 evaluation (values
within one rounding of the threshold).  One flipped ReLU mask moves a bias gradient by that
element's upstream gradient: 3e-4 of scale on fp2, 2e-2 on sa2 (tools/diag_fp2.py: 1 flip in
2 097 152, the kernel equals the sum over ITS mask to 1e-7).  So the float64 evaluation (c)
takes the discrete decisions -- every layer's ReLU mask and the pooled layer's arg-max --
from the fused run (they are saved for its backward), and everything continuous (products,
BatchNorm statistics and their backward, the gather / scatter, interpolation) is evaluated
in float64.  Against that the fused path must hold 2e-05 of every tensor's scale, forward
and every gradient; the same comparison with the decisions NOT forced is reported beside
it (op-by-op fp32 and fused both at 1e-6..2e-2 depending on how many thresholds flipped).
A missing or wrong term in a kernel shows up at 1e-2..1 whatever the decisions.
"""
import contextlib
import copy
import json
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

TOL = 2e-5            # fused fp32 path vs float64 with the same discrete decisions


def _e(a, t):
    a, t = a.detach().double(), t.detach().double()
    return float((a - t).abs().max() / max(1e-30, float(t.abs().max())))


@pytest.fixture(scope="module")
def captured():
    """One fused cfg3 train step; what every module saw and the gradient it got back."""
    from tests import test_configs_gpu as T
    from scan2cap_amd.loss_helper import get_scene_cap_loss
    from scan2cap_amd.pointnet2 import fused
    bench, wl, model, dd, batch, msa, dev = T._setup("cfg3")
    cfg = bench.LossConfig(msa)
    cap = {}
    undo = []

    def spy_sa(name, sa):
        orig = sa.forward

        def fwd(xyz, features=None, inds=None, geom=None):
            out = orig(xyz, features, inds=inds, geom=geom)
            rec = cap.setdefault(name, {})
            rec["xyz"], rec["feat"] = xyz.detach(), features.detach()
            rec["inds"] = out[2].detach().clone()
            rec["feat_needs_grad"] = bool(features.requires_grad)
            out[1].register_hook(lambda g: rec.__setitem__("dOut", g.detach().clone()))
            return out
        sa.forward = fwd
        undo.append(sa)

    def spy_fp(name, fp):
        orig = fp.forward

        def fwd(unknown, known, unknow_feats, known_feats, geom=None):
            out = orig(unknown, known, unknow_feats, known_feats, geom=geom)
            rec = cap.setdefault(name, {})
            rec.update(unknown=unknown.detach(), known=known.detach(),
                       uf=unknow_feats.detach().clone(), kf=known_feats.detach().clone())
            out.register_hook(lambda g: rec.__setitem__("dOut", g.detach().clone()))
            return out
        fp.forward = fwd
        undo.append(fp)

    bb = model.backbone_net
    for n in ("sa1", "sa2", "sa3", "sa4"):
        spy_sa(n, getattr(bb, n))
    spy_sa("vote", model.proposal.vote_aggregation)
    spy_fp("fp1", bb.fp1)
    spy_fp("fp2", bb.fp2)
    orig_rows = fused.mlp_rows

    def spy_rows(X, specs, params, pool_ns=0):
        out = orig_rows(X, specs, params, pool_ns)
        if X.shape == (wl["B"] * wl["K"], 128) and out.shape[1] == 97:
            rec = cap.setdefault("head", {})
            rec["X"] = X.detach().clone()
            out.register_hook(lambda g: rec.__setitem__("dOut", g.detach().clone()))
        return out
    fused.mlp_rows = spy_rows
    state = {k: v.clone() for k, v in model.state_dict().items()}
    try:
        d = model(dict(dd), use_tf=True, is_eval=False)
        for k in ("vote_xyz", "vote_features"):
            d[k].retain_grad()
        d = get_scene_cap_loss(d, dev, cfg, None, detection=True, caption=True,
                               orientation=False, distance=False)
        d["loss"].backward()
    finally:
        fused.mlp_rows = orig_rows
        for m in undo:
            del m.forward
    cap["vgen"] = dict(sx=d["seed_xyz"].detach().clone(),
                       sf=d["seed_features"].detach().clone().contiguous(),
                       gx=d["vote_xyz"].grad.clone(), gf=d["vote_features"].grad.clone().contiguous())
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    cap["_model"], cap["_wl"], cap["_pc"] = model, wl, dd["point_clouds"]
    return cap


def _judge(name, rows):
    """rows: key -> (op-by-op fp32 vs free float64, fused vs free float64,
    fused vs decision-forced float64); only the last one is asserted."""
    out = os.environ.get("S2C_GOLDEN_REPORT")
    if out:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "module_fp64_%s.json" % name), "w") as f:
            json.dump({"tol": TOL, "keys": {k: {"opbyop_vs_free": v[0], "fused_vs_free": v[1],
                                                "fused_vs_forced": v[2]}
                                            for k, v in rows.items()}}, f, indent=1)
    bad = {k: v[2] for k, v in rows.items() if not v[2] <= TOL}
    assert not bad, "%s: fused path beyond %.0e of scale vs float64 (same ReLU masks / " \
                    "arg-maxes): %s" % (name, TOL, bad)


def _find_ctx(fn, depth=0):
    """The _MLPRows autograd node behind an output (its ctx.saved = per-layer records)."""
    if fn is None or depth > 16:
        return None
    if isinstance(getattr(fn, "saved", None), list):
        return fn
    for nf, _ in fn.next_functions:
        r = _find_ctx(nf, depth + 1)
        if r is not None:
            return r
    return None


def _decisions(y):
    """Per layer of the fused run behind `y`: (ReLU mask (M,C) bool or None, arg (J,C) or None,
    pooled mask (J,C) or None) -- the discrete decisions its backward will use."""
    ctx = _find_ctx(y.grad_fn)
    assert ctx is not None, "no fused layer stack behind this output"
    out = []
    for r in ctx.saved:
        mask = arg = pmask = None
        if torch.is_tensor(r.get("arg")):
            arg = r["arg"].long().clone()
            pmask = (r["ymax"] * r["scale"] + r["shift"]) > 0
        elif torch.is_tensor(r.get("scale")) and r.get("relu"):
            mask = (r["Y"] * r["scale"] + r["shift"]) > 0
        elif r.get("relu"):
            mask = r["Y"] > 0
        out.append((mask, arg, pmask))
    return out


def _forced_stack64(X, layers, decisions, pool_ns=0):
    """float64 rows MLP: X (M,Cin) -> per layer Y = X W^T (+ b); train-mode BatchNorm (biased
    variance, pytorch_utils.py:100-120 -> nn.BatchNorm2d); ReLU as multiplication with the
    GIVEN mask; last layer with pool_ns: the GIVEN arg-max row of every (centre, channel)."""
    A = X
    for (W, bias, bn), (mask, arg, pmask) in zip(layers, decisions):
        Y = A @ W.view(W.shape[0], -1).t()
        if bias is not None:
            Y = Y + bias
        if bn is not None:
            mu, var = Y.mean(0), Y.var(0, unbiased=False)
            Y = (Y - mu) / torch.sqrt(var + bn.eps) * bn.weight + bn.bias
        if arg is not None:
            J = Y.shape[0] // pool_ns
            A = Y.view(J, pool_ns, -1).gather(1, arg.unsqueeze(1)).squeeze(1) * pmask
        elif mask is not None:
            A = Y * mask
        else:
            A = Y
    return A


def _mlp_layers64(mlp64):
    return [(layer.conv.weight, layer.conv.bias, layer.bn.bn if hasattr(layer, "bn") else None)
            for layer in mlp64.children()]


def _sa_three_ways(rec, sa, pc=None):
    """pc: for SA1 the (B,N,3+C) cloud -- the fused path must read its 540-byte rows in
    place (features arrive as the transposed VIEW of the cloud, backbone_module.py:68-72)."""
    from scan2cap_amd.opbyop import op_by_op
    from scan2cap_amd.pointnet2 import _ext
    xyz, inds, dOut = rec["xyz"], rec["inds"], rec["dOut"].contiguous()
    needs = rec["feat_needs_grad"]

    def run(ctx):
        mod = copy.deepcopy(sa).train()
        if pc is not None:
            f = pc[..., 3:].transpose(1, 2)
            assert f.stride() == rec["feat"].stride() and not f.is_contiguous()
        else:
            f = rec["feat"].clone().requires_grad_(needs)
        x = xyz.clone().requires_grad_(needs)
        with ctx:
            nx, nf, ni = mod(x, f, inds=inds)
        assert torch.equal(ni, inds)
        dec = _decisions(nf) if fused_run else None
        (nf * dOut).sum().backward()
        g = {n: p.grad for n, p in mod.named_parameters()}
        return g, (f.grad if needs else None), (x.grad if needs else None), nf.detach(), dec

    fused_run = True
    fused_out = run(contextlib.nullcontext())
    fused_run = False
    ref_out = run(op_by_op())
    # float64: QueryAndGroup -> 3 x (1x1 conv, BatchNorm2d, ReLU) -> max over nsample
    new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3))
    idx = _ext.ball_query(new_xyz.contiguous(), xyz.contiguous(), sa.radius, sa.nsample).long()
    B, m, ns = idx.shape

    def grp(t):                                   # (B,C,N) -> (B,C,m,ns)
        return torch.gather(t, 2, idx.view(B, 1, m * ns).expand(-1, t.shape[1], -1)).view(
            B, t.shape[1], m, ns)

    def eval64(decisions):
        x64 = xyz.double().requires_grad_(needs)
        f64 = rec["feat"].double().contiguous().requires_grad_(needs)
        mod64 = copy.deepcopy(sa).double().train()
        centre = torch.gather(x64, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3))
        gx = (grp(x64.transpose(1, 2)) - centre.transpose(1, 2).unsqueeze(-1)) / sa.radius
        h = torch.cat([gx, grp(f64)], 1)                           # pointnet2_utils.py:348-359
        if decisions is None:
            y64 = F.max_pool2d(mod64.mlp_module(h), kernel_size=[1, ns]).squeeze(-1)
        else:        # rows (b, centre, sample) x channels, the fused run's masks and arg-maxes
            X = h.permute(0, 2, 3, 1).reshape(B * m * ns, -1)
            y64 = _forced_stack64(X, _mlp_layers64(mod64.mlp_module), decisions, ns)
            y64 = y64.view(B, m, -1).transpose(1, 2)
        (y64 * dOut.double()).sum().backward()
        return ({n: p.grad for n, p in mod64.named_parameters()},
                f64.grad if needs else None, x64.grad if needs else None, y64.detach())
    free, forced = eval64(None), eval64(fused_out[4])
    rows = {"forward": (_e(ref_out[3], free[3]), _e(fused_out[3], free[3]),
                        _e(fused_out[3], forced[3]))}
    if needs:
        rows["d features"] = (_e(ref_out[1], free[1]), _e(fused_out[1], free[1]),
                              _e(fused_out[1], forced[1]))
        rows["d xyz"] = (_e(ref_out[2], free[2]), _e(fused_out[2], free[2]),
                         _e(fused_out[2], forced[2]))
    for n in free[0]:
        rows[n] = (_e(ref_out[0][n], free[0][n]), _e(fused_out[0][n], free[0][n]),
                   _e(fused_out[0][n], forced[0][n]))
    return rows


@pytest.mark.parametrize("name", ["sa1", "sa2", "sa3", "sa4", "vote"])
def test_set_abstraction_stage_vs_float64(captured, name):
    model = captured["_model"]
    sa = model.proposal.vote_aggregation if name == "vote" else getattr(model.backbone_net, name)
    rec = captured[name]
    if name == "sa1":
        B, C, N = rec["feat"].shape
        assert (B * sa.npoint * sa.nsample, C, N) == (1048576, 132, 40000)
        assert rec["feat"].stride(2) * 4 == 540            # (B,N,135) rows read in place
    rows = _sa_three_ways(rec, sa, pc=captured["_pc"] if name == "sa1" else None)
    _judge(name, rows)


@pytest.mark.parametrize("name", ["fp1", "fp2"])
def test_feature_propagation_vs_float64(captured, name):
    from scan2cap_amd.opbyop import op_by_op
    from scan2cap_amd.pointnet2.pointnet2_modules import PointnetFPModule
    fp = getattr(captured["_model"].backbone_net, name)
    rec = captured[name]
    dOut = rec["dOut"].contiguous()
    geom = PointnetFPModule.geometry(rec["unknown"], rec["known"])

    def run(ctx):
        mod = copy.deepcopy(fp).train()
        uf = rec["uf"].clone().requires_grad_(True)
        kf = rec["kf"].clone().requires_grad_(True)
        with ctx:
            y = mod(rec["unknown"], rec["known"], uf, kf, geom=geom)
        dec = _decisions(y) if fused_run else None
        (y * dOut).sum().backward()
        return ({n: p.grad for n, p in mod.named_parameters()}, uf.grad, kf.grad, y.detach(),
                dec)
    fused_run = True
    fo = run(contextlib.nullcontext())
    fused_run = False
    ro = run(op_by_op())
    idx, w = geom[0].long(), geom[1].double()
    B, n, _ = idx.shape

    def eval64(decisions):
        uf = rec["uf"].double().requires_grad_(True)
        kf = rec["kf"].double().requires_grad_(True)
        mod64 = copy.deepcopy(fp).double().train()
        g = torch.gather(kf, 2, idx.view(B, 1, n * 3).expand(-1, kf.shape[1], -1)).view(
            B, kf.shape[1], n, 3)
        interp = (g * w.unsqueeze(1)).sum(-1)                     # interpolate_gpu.cu:87-99
        h = torch.cat([interp, uf], 1)                            # pointnet2_modules.py:408
        if decisions is None:
            y64 = mod64.mlp(h.unsqueeze(-1)).squeeze(-1)
        else:
            X = h.transpose(1, 2).reshape(B * n, -1)
            y64 = _forced_stack64(X, _mlp_layers64(mod64.mlp), decisions).view(
                B, n, -1).transpose(1, 2)
        (y64 * dOut.double()).sum().backward()
        return ({nme: p.grad for nme, p in mod64.named_parameters()}, uf.grad, kf.grad,
                y64.detach())
    free, forced = eval64(None), eval64(fo[4])
    rows = {"forward": (_e(ro[3], free[3]), _e(fo[3], free[3]), _e(fo[3], forced[3])),
            "d unknown feats": (_e(ro[1], free[1]), _e(fo[1], free[1]), _e(fo[1], forced[1])),
            "d known feats": (_e(ro[2], free[2]), _e(fo[2], free[2]), _e(fo[2], forced[2]))}
    for nme in free[0]:
        rows[nme] = (_e(ro[0][nme], free[0][nme]), _e(fo[0][nme], free[0][nme]),
                     _e(fo[0][nme], forced[0][nme]))
    _judge(name, rows)


def test_voting_module_vs_float64(captured):
    from scan2cap_amd.opbyop import op_by_op
    rec, vgen = captured["vgen"], captured["_model"].vgen
    dec = []

    def run(ctx, dtype=torch.float32):
        mod = copy.deepcopy(vgen).to(dtype).train()
        x = rec["sx"].to(dtype).clone().requires_grad_(True)
        f = rec["sf"].to(dtype).clone().requires_grad_(True)
        with ctx:
            if dtype == torch.float32 and hasattr(mod, "forward_normalized"):
                vx, vf = mod.forward_normalized(x, f)
                if _find_ctx(vf.grad_fn) is not None:
                    dec.append(_decisions(vf))
            else:
                vx, vf = mod(x, f)
                vf = vf.div(torch.norm(vf, p=2, dim=1).unsqueeze(1))   # capnet.py:97-98
        ((vx * rec["gx"].to(dtype)).sum() + (vf * rec["gf"].to(dtype)).sum()).backward()
        return ({n: p.grad for n, p in mod.named_parameters()}, f.grad, x.grad,
                vf.detach(), vx.detach())
    fo = run(contextlib.nullcontext())
    assert len(dec) == 1, "the fused vote MLP did not run"
    ro = run(op_by_op())
    with op_by_op():
        t = run(contextlib.nullcontext(), torch.float64)

    def forced64(decisions):
        """float64 with the fused run's ReLU masks (8192 x 256 thresholds per layer: ONE pre-
        activation within float32 rounding of zero flips a mask and moves a weight gradient by
        1e-4..1e-3 of scale -- seen on some boxes, not on others, with the free evaluation)"""
        mod = copy.deepcopy(vgen).double().train()
        x = rec["sx"].double().clone().requires_grad_(True)
        f = rec["sf"].double().clone().requires_grad_(True)
        B, S, C = x.shape[0], x.shape[1], f.shape[1]
        rows = f.transpose(2, 1).reshape(B * S, C)
        net = _forced_stack64(rows, [(mod.conv1.weight, mod.conv1.bias, mod.bn1),
                                     (mod.conv2.weight, mod.conv2.bias, mod.bn2),
                                     (mod.conv3.weight, mod.conv3.bias, None)], decisions)
        vx = (x.reshape(B * S, 3) + net[:, :3]).view(B, S, 3)              # voting_module.py:50-57
        vf = rows + net[:, 3:]
        vf = (vf / vf.norm(p=2, dim=1, keepdim=True)).view(B, S, C).transpose(2, 1)   # capnet.py:97-98
        ((vx * rec["gx"].double()).sum() + (vf * rec["gf"].double()).sum()).backward()
        return ({n: p.grad for n, p in mod.named_parameters()}, f.grad, x.grad,
                vf.detach(), vx.detach())
    d = forced64(dec[0])
    rows = {"vote_features": (_e(ro[3], t[3]), _e(fo[3], t[3]), _e(fo[3], d[3])),
            "vote_xyz": (_e(ro[4], t[4]), _e(fo[4], t[4]), _e(fo[4], d[4])),
            "d seed_features": (_e(ro[1], t[1]), _e(fo[1], t[1]), _e(fo[1], d[1]))}
    for n, g in d[0].items():
        if float(g.abs().max()) < 1e-9:     # conv biases in front of a BatchNorm: exactly 0
            assert float(fo[0][n].abs().max()) < 1e-5, n
            continue
        rows[n] = (_e(ro[0][n], t[0][n]), _e(fo[0][n], t[0][n]), _e(fo[0][n], g))
    _judge("vgen", rows)


def test_proposal_head_vs_float64(captured):
    from scan2cap_amd.pointnet2 import fused
    wl, rec = captured["_wl"], captured["head"]
    p = captured["_model"].proposal.proposal
    X, dOut = rec["X"], rec["dOut"]
    B, K = wl["B"], wl["K"]

    def run_torch(dtype):
        seq = copy.deepcopy(p).to(dtype).train()
        x = X.to(dtype).clone().requires_grad_(True)
        y = seq(x.view(B, K, 128).transpose(1, 2))                      # proposal_module.py:46-54
        y.backward(dOut.to(dtype).view(B, K, -1).transpose(1, 2))
        return ({n: q.grad for n, q in seq.named_parameters()}, x.grad,
                y.transpose(1, 2).reshape(B * K, -1).detach())

    def run_fused():
        seq = copy.deepcopy(p).train()
        specs = [fused.LayerSpec(False, seq[1], True), fused.LayerSpec(False, seq[4], True),
                 fused.LayerSpec(True, None, False)]
        params = [seq[0].weight.view(128, -1), seq[1].weight, seq[1].bias,
                  seq[3].weight.view(128, -1), seq[4].weight, seq[4].bias,
                  seq[6].weight.view(seq[6].out_channels, -1), seq[6].bias]
        x = X.clone().requires_grad_(True)
        y = fused.mlp_rows(x, specs, params)
        dec = _decisions(y)
        y.backward(dOut)
        return {n: q.grad for n, q in seq.named_parameters()}, x.grad, y.detach(), dec

    def run_forced(decisions):
        seq = copy.deepcopy(p).double().train()
        x = X.double().requires_grad_(True)
        layers = [(seq[0].weight, None, seq[1]), (seq[3].weight, None, seq[4]),
                  (seq[6].weight, seq[6].bias, None)]
        y = _forced_stack64(x, layers, decisions)
        y.backward(dOut.double())
        return {n: q.grad for n, q in seq.named_parameters()}, x.grad, y.detach()
    t, ro, fo = run_torch(torch.float64), run_torch(torch.float32), run_fused()
    fr = run_forced(fo[3])
    rows = {"forward": (_e(ro[2], t[2]), _e(fo[2], t[2]), _e(fo[2], fr[2])),
            "d X": (_e(ro[1], t[1]), _e(fo[1], t[1]), _e(fo[1], fr[1]))}
    for n, g in t[0].items():
        rows[n] = (_e(ro[0][n], g), _e(fo[0][n], g), _e(fo[0][n], fr[0][n]))
    _judge("head", rows)
