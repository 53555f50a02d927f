"""Which fused piece moves the cfg3 gradients away from the op-by-op run by more than the
noise level?  Reference: all op-by-op.  Variants: exactly one piece fused."""
import os, sys, importlib, contextlib
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_configs_gpu as T
from tests import golden_common as gc
from scan2cap_amd.loss_helper import get_scene_cap_loss
from scan2cap_amd import opbyop
from scan2cap_amd.pointnet2 import fused, pointnet2_modules as pm
bench, wl, model, dd, batch, msa, dev = T._setup("cfg3")
cfg = bench.LossConfig(msa)
state = {k: v.clone() for k, v in model.state_dict().items()}
KEYS = ("sa4_features", "fp2_features", "vote_xyz", "vote_features")
def run():
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    d = model(dict(dd), use_tf=True, is_eval=False)
    for k in KEYS:
        d[k].retain_grad()
    d = get_scene_cap_loss(d, dev, cfg, None, detection=True, caption=True, orientation=False, distance=False)
    d["loss"].backward()
    g = {k: d[k].grad.clone() for k in KEYS}
    g.update({n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None and n in (
        "backbone_net.sa4.mlp_module.layer0.conv.weight", "backbone_net.fp2.mlp.layer0.conv.weight", "vgen.conv2.weight",
        "proposal.vote_aggregation.mlp_module.layer1.conv.weight", "backbone_net.sa1.mlp_module.layer0.conv.weight")})
    return d, g

F, gF = run()
forced = F["aggregated_vote_inds"]

@contextlib.contextmanager
def flags(on):
    """all op-by-op except the (module, name) pairs in `on`"""
    saved = []
    for mod, name in opbyop._FLAGS:
        m = importlib.import_module(mod)
        saved.append((m, name, getattr(m, name)))
        setattr(m, name, (mod.split(".")[-1], name) in on)
    try:
        yield
    finally:
        for m, name, v in saved:
            setattr(m, name, v)

def rel(a, b): return float((a - b).abs().max() / b.abs().max())
with gc.forced_vote_sampling(model, forced):
    with flags(set()):
        O, gO = run()
        with gc.ulp_noise(model, 3):
            N, gN = run()
    print("%-34s" % "variant", " ".join("%-10s" % k[-22:-12] if len(k) > 22 else "%-10s" % k[:10] for k in gO))
    print("%-34s" % "noise probe", " ".join("%-10.2e" % rel(gN[k], gO[k]) for k in gO))
    print("%-34s" % "all fused", " ".join("%-10.2e" % rel(gF[k], gO[k]) for k in gO))
    pc0 = dd["point_clouds"]
    for eps in (1e-7, 3e-8):
        g = torch.Generator(device=dev).manual_seed(11)
        pc = pc0.clone()
        pc[..., 3:] *= (1 + eps * torch.randn(pc[..., 3:].shape, generator=g, device=dev))
        dd["point_clouds"] = pc
        with flags(set()):
            V, gV = run()
        dd["point_clouds"] = pc0
        print("%-34s" % ("op-by-op, feature inputs x(1+%.0e n)" % eps), " ".join("%-10.2e" % rel(gV[k], gO[k]) for k in gO))
    fused.set_gemm_split(False)
    V, gV = run()
    fused.set_gemm_split(True)
    print("%-34s" % "all fused, exact fp32 MFMA chain", " ".join("%-10.2e" % rel(gV[k], gO[k]) for k in gO))
    for k in ("sa1_features", "sa2_features"):
        a_, b_, c_ = V[k].double(), F[k].double(), O[k].double()
        print(k, "fp32chain-opbyop rms %.2e | split-opbyop rms %.2e | max %.2e %.2e" % (
            float((a_ - c_).pow(2).mean().sqrt() / c_.pow(2).mean().sqrt()), float((b_ - c_).pow(2).mean().sqrt() / c_.pow(2).mean().sqrt()),
            float((a_ - c_).abs().max() / c_.abs().max()), float((b_ - c_).abs().max() / c_.abs().max())))
