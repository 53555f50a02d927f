"""Two arithmetic variants of the train step on a golden fixture, side by side: forward tensors,
gradients of the intermediates, and the saved decisions (ReLU masks, pooled arg-max rows, BatchNorm
statistics) of every rows stack.  usage: diag_golden_ab.py <fixture> "<ENV=V ...>" "<ENV=V ...>" """
import os, sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests import golden_common as gc
from tests.test_capnet_golden import build_model, load_fixture
from scan2cap_amd.loss_helper import get_scene_cap_loss
from scan2cap_amd.pointnet2 import fused

name, envs = sys.argv[1], [dict(kv.split("=") for kv in a.split()) for a in sys.argv[2:4]]
spec, ref, inputs = load_fixture(name)
SAVED, CUR = {}, [0]
_bwd = fused._MLPRows.backward
def _bwd_rec(ctx, dOut):
    recs = [{k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in r.items()
             if k in ("Y", "arg", "ymax", "scale", "shift", "mean", "invstd")} for r in ctx.saved]
    SAVED.setdefault(CUR[0], []).append((recs, dOut.detach().clone()))
    return _bwd(ctx, dOut)
fused._MLPRows.backward = staticmethod(_bwd_rec)
FLAGS = {"S2C_POINT_SPACE": "POINT_SPACE"}    # (the S2C_POINT_GEMM_* variants were dropped in round 5)
res = []
for i, env in enumerate(envs):
    CUR[0] = i
    fused.POINT_SPACE = env.get("S2C_POINT_SPACE", "1") != "0"
    model, sd = build_model("cuda", name)
    model.train(); model.zero_grad()
    with gc.forced_vote_sampling(model, torch.from_numpy(ref["train/aggregated_vote_inds"])):
        dd = model(gc.to_torch(inputs, "cuda"), use_tf=True, is_eval=False)
    keep = {}
    for k, v in dd.items():
        if torch.is_tensor(v) and v.is_floating_point() and v.requires_grad and not v.is_leaf:
            v.retain_grad(); keep[k] = v
    ints = {k: v.detach().clone() for k, v in dd.items() if torch.is_tensor(v) and not v.is_floating_point()}
    dd = get_scene_cap_loss(dd, torch.device("cuda"), gc.LossConfig(gc.mean_size_arr()), None, **gc.LOSS_FLAGS)
    dd["loss"].backward()
    res.append(({k: v.detach().clone() for k, v in keep.items()},
                {k: v.grad.clone() for k, v in keep.items() if v.grad is not None}, gc.extract_grads(model), ints))
def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
a, b = res
for k in a[3]:
    if a[3][k].shape == b[3][k].shape and int((a[3][k] != b[3][k]).sum()):
        print("INTEGER key %s: %d mismatches" % (k, int((a[3][k] != b[3][k]).sum())))
print("%-32s forward  gradient" % "intermediate")
for k in a[0]:
    print("  %-30s %.2e %s" % (k, rel(a[0][k], b[0][k]), ("%.2e" % rel(a[1][k], b[1][k])) if k in a[1] and k in b[1] else "-"))
print("saved decisions per rows stack (backward order)")
for i, ((ra, da), (rb, db)) in enumerate(zip(SAVED[0], SAVED[1])):
    print(" stack %d: dOut diff %.2e rows %s" % (i, rel(da, db), [tuple(r["Y"].shape) if r.get("Y") is not None else None for r in ra]))
    for li, (x, y) in enumerate(zip(ra, rb)):
        msg = []
        if x.get("arg") is not None:
            msg.append("arg mismatches %d/%d" % (int((x["arg"] != y["arg"]).sum()), x["arg"].numel()))
        if x.get("Y") is not None and x.get("scale") is not None:
            ma = (x["Y"] * x["scale"] + x["shift"]) > 0
            mb = (y["Y"] * y["scale"] + y["shift"]) > 0
            msg.append("relu mask mismatches %d/%d" % (int((ma != mb).sum()), ma.numel()))
        if x.get("invstd") is not None:
            msg.append("invstd max %.2e (rel diff %.1e)" % (float(x["invstd"].max()), rel(x["invstd"], y["invstd"])))
        if msg:
            print("    layer %d: %s" % (li, "; ".join(msg)))
rows = sorted(((rel(torch.from_numpy(a[2][k]), torch.from_numpy(b[2][k])), k) for k in a[2]), reverse=True)
print("parameter gradients, largest differences:", ["%s %.1e" % (k, r) for r, k in rows[:6]])
