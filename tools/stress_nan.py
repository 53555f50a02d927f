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
    # (the eager reference below runs on a side stream too: a default-stream backward before a
    # capture binds AccumulateGrad nodes to the legacy stream and crashes the capture)
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
# the replayed graph writes its gradients into the tensors p.grad pointed at when the capture ended
gg = {n: p.grad for n, p in model.named_parameters() if p.grad is not None} if mode != "eager" else None
# eager reference gradients from the same weights (graph replays must reproduce them up to the
# last-bit noise of the float atomics: a mis-ordered memset / memcpy node shows up here even
# when the garbage happens to be finite)
model.load_state_dict(state)
float(step().detach())
ref = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
worst = 0.0
for i in range(iters):
    model.load_state_dict(state)
    loss = float(fn().detach())
    cur = gg if gg is not None else {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    for n, g in cur.items():
        if n in ref:
            e = float((g - ref[n]).abs().max()) / max(1.0, float(ref[n].abs().max()))
            if e == e:
                worst = max(worst, e)
    ng = [n for n, g in cur.items() if not torch.isfinite(g).all()]
    if ng:
        bad += 1
        if first is None:
            first = (i, loss, len(ng), ng[:2], ng[-2:])
print(mode, sys.argv[3:] , "steps", iters, "steps with non-finite grads:", bad, "first:", first,
      "| worst gradient deviation from the eager step: %.2e of scale" % worst)
