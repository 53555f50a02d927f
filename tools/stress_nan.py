"""Stress: repeat ONE training step from identical weights many times and look for
non-finite gradients (an intermittent NaN was seen in the SA backward)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_checkpoint as T
from scan2cap_amd.loss_helper import get_scene_cap_loss
from scan2cap_amd.pointnet2 import _ext
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
if len(sys.argv) > 3 and sys.argv[3] == "brute":
    _ext.BQ_GRID_MIN_N = 10**9
bench, wl, model, dd, cfg, dev = T._train_setup()
state = {k: v.clone() for k, v in model.state_dict().items()}
def step():
    model.zero_grad(set_to_none=True)
    d = model(dict(dd), use_tf=True, is_eval=False)
    d = get_scene_cap_loss(d, dev, cfg, None, detection=True, caption=True, orientation=False, distance=False)
    d["loss"].backward()
    return d["loss"]
if mode == "graph":
    from scan2cap_amd.graphs import GraphedCallable
    g = GraphedCallable(step).capture()
    fn = g
elif mode == "slots":
    from scan2cap_amd.graphs import GraphedCallable
    from scan2cap_amd.pipeline import GeometrySlots
    slots = GeometrySlots(model.backbone_net, dd["point_clouds"], 1)
    dd["_geometry"] = slots.geometry(0)
    g = GraphedCallable(step).capture()
    slots.refill(0, dd["point_clouds"])
    slots.acquire(0)
    fn = g
else:
    fn = step
bad = 0
first = None
for i in range(iters):
    model.load_state_dict(state)
    loss = float(fn().detach())
    if i < 14:
        print(i, "loss %.6f" % loss, "nan-grad params", sum(1 for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()))
    ng = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    if ng:
        bad += 1
        if first is None:
            first = (i, loss, len(ng), ng[:2], ng[-2:])
print(mode, sys.argv[3:] , "steps", iters, "steps with non-finite grads:", bad, "first:", first)
