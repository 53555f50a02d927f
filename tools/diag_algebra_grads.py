"""Gradients of one train step with the pooled-layer algebra on / off, and run-to-run."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_train_loop_gpu as T
from scan2cap_amd.pointnet2 import fused
from scan2cap_amd.loss_helper import get_scene_cap_loss
bench, wl, model, opt, dd, cfg, dev = T._setup()
def grads(on):
    fused.POOL_ALGEBRA = on
    model.zero_grad(set_to_none=True)
    d = model(dict(dd), use_tf=True, is_eval=False)
    d = get_scene_cap_loss(d, dev, cfg, None)
    d["loss"].backward()
    return float(d["loss"].detach()), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
state = {k: v.clone() for k, v in model.state_dict().items()}
runs = []
for on in (True, True, False, False):
    model.load_state_dict(state)
    runs.append(grads(on))
print("loss", [r[0] for r in runs])
def cmp(a, b, label):
    rows = []
    for n in a:
        s = max(1e-12, float(b[n].abs().max()))
        rows.append((float((a[n] - b[n]).abs().max()) / s, n, s))
    rows.sort(reverse=True)
    print(label, "worst:", ["%s %.1e" % (r[1][-40:], r[0]) for r in rows[:6]])
cmp(runs[0][1], runs[1][1], "on vs on ")
cmp(runs[2][1], runs[3][1], "off vs off")
cmp(runs[0][1], runs[2][1], "on vs off")
