import copy, os, sys, contextlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_configs_gpu as T
from scan2cap_amd.loss_helper import get_scene_cap_loss
from scan2cap_amd.opbyop import op_by_op
bench, wl, model, dd, batch, msa, dev = T._setup("cfg3")
cfg = bench.LossConfig(msa)
d = model(dict(dd), use_tf=True, is_eval=False)
for k in ("vote_xyz", "vote_features"):
    d[k].retain_grad()
d = get_scene_cap_loss(d, dev, cfg, None, detection=True, caption=True, orientation=False, distance=False)
d["loss"].backward()
sx, sf = d["seed_xyz"].detach().clone(), d["seed_features"].detach().clone().contiguous()
gx, gf = d["vote_xyz"].grad.clone(), d["vote_features"].grad.clone().contiguous()
print("seed_features", tuple(sf.shape), "gx %.3e gf %.3e" % (gx.abs().max(), gf.abs().max()))
def run(ctx, dtype=torch.float32):
    mod = copy.deepcopy(model.vgen).to(dtype).train()
    x = sx.to(dtype).clone().requires_grad_(True)
    f = sf.to(dtype).clone().requires_grad_(True)
    with ctx:
        if dtype == torch.float32 and hasattr(mod, "forward_normalized"):
            vx, vf = mod.forward_normalized(x, f)
        else:
            vx, vf = mod(x, f)
            vf = vf.div(torch.norm(vf, p=2, dim=1).unsqueeze(1))
    ((vx * gx.to(dtype)).sum() + (vf * gf.to(dtype)).sum()).backward()
    return {n: p.grad.double() for n, p in mod.named_parameters()}, f.grad.double(), x.grad.double(), vf.detach().double(), vx.detach().double()
F = run(contextlib.nullcontext())
O = run(op_by_op())
with op_by_op():
    Tt = run(contextlib.nullcontext(), torch.float64)
def e(a, t): return float((a - t).abs().max() / max(1e-30, float(t.abs().max())))
print("vote_features fwd: opbyop %.2e fused %.2e ; vote_xyz: %.2e %.2e" % (e(O[3], Tt[3]), e(F[3], Tt[3]), e(O[4], Tt[4]), e(F[4], Tt[4])))
print("d seed_features : opbyop %.2e fused %.2e" % (e(O[1], Tt[1]), e(F[1], Tt[1])))
print("d seed_xyz      : opbyop %.2e fused %.2e" % (e(O[2], Tt[2]), e(F[2], Tt[2])))
for n in Tt[0]:
    print("%-14s max %.3e opbyop %.2e fused %.2e" % (n, Tt[0][n].abs().max(), e(O[0][n], Tt[0][n]), e(F[0][n], Tt[0][n])))
