"""Isolate one SA stage of the cfg3 model (default: the vote aggregation): the inputs and
the upstream gradient of the fused full run, then (a) fused fp32, (b) op-by-op fp32,
(c) a float64 torch emulation of the op-by-op formulation, all from IDENTICAL inputs."""
import copy, os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_configs_gpu as T
from scan2cap_amd.loss_helper import get_scene_cap_loss
from scan2cap_amd.opbyop import op_by_op

which = sys.argv[1] if len(sys.argv) > 1 else "vote"
bench, wl, model, dd, batch, msa, dev = T._setup("cfg3")
cfg = bench.LossConfig(msa)
sa = {"vote": model.proposal.vote_aggregation, "sa4": model.backbone_net.sa4,
      "sa2": model.backbone_net.sa2}[which]
cap = {}
orig = sa.forward
def spy(xyz, features=None, inds=None, geom=None):
    out = orig(xyz, features, inds=inds, geom=geom)
    cap["xyz"], cap["feat"] = xyz.detach().clone(), features.detach().clone().contiguous()
    cap["inds"] = out[2].detach().clone()
    out[1].register_hook(lambda g: cap.__setitem__("dOut", g.detach().clone()))
    return out
sa.forward = spy
state = {k: v.clone() for k, v in model.state_dict().items()}
d = model(dict(dd), use_tf=True, is_eval=False)
d = get_scene_cap_loss(d, dev, cfg, None, detection=True, caption=True, orientation=False, distance=False)
d["loss"].backward()
del sa.forward
model.load_state_dict(state)
xyz, feat, inds, dOut = cap["xyz"], cap["feat"], cap["inds"], cap["dOut"].contiguous()
print("xyz", tuple(xyz.shape), "feat", tuple(feat.shape), "absmax %.3f" % feat.abs().max(), "dOut", tuple(dOut.shape), "%.3e" % dOut.abs().max())

def run(mod, ctx):
    mod = copy.deepcopy(mod).train()
    f = feat.clone().requires_grad_(True)
    x = xyz.clone().requires_grad_(True)
    with ctx:
        nx, nf, ni = mod(x, f, inds=inds)
    assert torch.equal(ni, inds)
    (nf * dOut).sum().backward()
    return ({n: p.grad.double() for n, p in mod.named_parameters()}, f.grad.double(), x.grad.double(), nf.detach().double())

import contextlib
gf, dff, dxf, yf = run(sa, contextlib.nullcontext())
go, dfo, dxo, yo = run(sa, op_by_op())

# float64 emulation (QueryAndGroup -> conv/BN/ReLU x3 -> max), same ball-query idx
from scan2cap_amd.pointnet2 import _ext
new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3))
idx = _ext.ball_query(new_xyz.contiguous(), xyz.contiguous(), sa.radius, sa.nsample).long()
B, m, ns = idx.shape
x64 = xyz.double().requires_grad_(True)
f64 = feat.double().requires_grad_(True)
mod64 = copy.deepcopy(sa).double().train()
def grp(t):   # t (B,C,N) -> (B,C,m,ns)
    return torch.gather(t, 2, idx.view(B, 1, m * ns).expand(-1, t.shape[1], -1)).view(B, t.shape[1], m, ns)
gx = grp(x64.transpose(1, 2)) - torch.gather(x64, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).transpose(1, 2).unsqueeze(-1)
gx = gx / sa.radius
h = torch.cat([gx, grp(f64)], 1)
h = mod64.mlp_module(h)
y64 = F.max_pool2d(h, kernel_size=[1, ns]).squeeze(-1)
(y64 * dOut.double()).sum().backward()
gt = {n: p.grad for n, p in mod64.named_parameters()}
def e(a, t): return float((a - t).abs().max() / max(1e-30, float(t.abs().max())))
print("forward : opbyop %.2e fused %.2e" % (e(yo, y64.detach()), e(yf, y64.detach())))
print("d feat  : opbyop %.2e fused %.2e" % (e(dfo, f64.grad), e(dff, f64.grad)))
print("d xyz   : opbyop %.2e fused %.2e" % (e(dxo, x64.grad), e(dxf, x64.grad)))
for n in gt:
    print("%-34s max %.3e  opbyop %.2e  fused %.2e" % (n, gt[n].abs().max(), e(go[n], gt[n]), e(gf[n], gt[n])))
