"""Isolate the proposal head MLP at cfg3: same input rows + same upstream gradient through
(a) fused.mlp_rows fp32, (b) torch Sequential fp32, (c) torch Sequential fp64."""
import copy, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_configs_gpu as T
from scan2cap_amd.loss_helper import get_scene_cap_loss
from scan2cap_amd.pointnet2 import fused
from scan2cap_amd.models import proposal_module

bench, wl, model, dd, batch, msa, dev = T._setup("cfg3")
cfg = bench.LossConfig(msa)
cap = {}
orig = fused.mlp_rows
def spy(X, specs, params, pool_ns=0):
    out = orig(X, specs, params, pool_ns)
    if X.shape == (wl["B"] * wl["K"], 128) and out.shape[1] == 97:
        cap["X"] = X.detach().clone()
        out.register_hook(lambda g: cap.__setitem__("dOut", g.detach().clone()))
    return out
fused.mlp_rows = spy
state = {k: v.clone() for k, v in model.state_dict().items()}
d = model(dict(dd), use_tf=True, is_eval=False)
d = get_scene_cap_loss(d, dev, cfg, None, detection=True, caption=True, orientation=False, distance=False)
d["loss"].backward()
fused.mlp_rows = orig
model.load_state_dict(state)
X, dOut = cap["X"], cap["dOut"]
print("X", X.shape, "mean %.3f std %.3f min %.3f frac0 %.3f" % (X.mean(), X.std(), X.min(), (X == 0).float().mean()))
print("dOut absmax %.3e" % dOut.abs().max())
p = model.proposal.proposal

def run_torch(seq, x, g, dtype):
    seq = copy.deepcopy(seq).to(dtype).train()
    x = x.to(dtype).clone().requires_grad_(True)
    B, K = wl["B"], wl["K"]
    y = seq(x.view(B, K, 128).transpose(1, 2))             # (B,C,K)
    y.backward(g.to(dtype).view(B, K, -1).transpose(1, 2))
    return {n: q.grad.double() for n, q in seq.named_parameters()}, x.grad.double(), y.transpose(1, 2).reshape(B * K, -1).double(), seq

def run_fused(x, g):
    seq = copy.deepcopy(p).train()
    specs = [fused.LayerSpec(False, seq[1], True), fused.LayerSpec(False, seq[4], True), fused.LayerSpec(True, None, False)]
    params = [seq[0].weight.view(128, -1), seq[1].weight, seq[1].bias, seq[3].weight.view(128, -1), seq[4].weight, seq[4].bias,
              seq[6].weight.view(seq[6].out_channels, -1), seq[6].bias]
    x = x.clone().requires_grad_(True)
    y = fused.mlp_rows(x, specs, params)
    y.backward(g)
    return {n: q.grad.double() for n, q in seq.named_parameters()}, x.grad.double(), y.double(), seq

gt, dxt, yt, s64 = run_torch(p, X, dOut, torch.float64)
go, dxo, yo, s32 = run_torch(p, X, dOut, torch.float32)
gf, dxf, yf, sf = run_fused(X, dOut)
def e(a, t): return float((a - t).abs().max() / max(1e-30, t.abs().max()))
print("forward err: torch32 %.2e fused %.2e" % (e(yo, yt), e(yf, yt)))
print("dX err:      torch32 %.2e fused %.2e" % (e(dxo, dxt), e(dxf, dxt)))
for n in gt:
    print("%-12s max %.3e  torch32 %.2e  fused %.2e" % (n, gt[n].abs().max(), e(go[n], gt[n]), e(gf[n], gt[n])))
# conditioning of the BN layers: |mean|/std of pre-BN activations
with torch.no_grad():
    Y0 = X.double() @ s64[0].weight.view(128, -1).t()
    print("layer0 pre-BN |mean|/std: max %.1f median %.1f" % ((Y0.mean(0).abs() / Y0.std(0)).max(), (Y0.mean(0).abs() / Y0.std(0)).median()))
    for nm, bn in (("bn1 fused", sf[1]), ("bn1 torch32", s32[1]), ("bn1 fp64", s64[1])):
        print(nm, "running_mean[:4]", bn.running_mean[:4].tolist(), "running_var[:4]", bn.running_var[:4].tolist())
