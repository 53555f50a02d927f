"""Overfit one fixed cfg3 batch with the captured step (hipGraph replay + geometry slots): the
loss must fall.  End-to-end sanity of every backward kernel on the replay path."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_configs_gpu as T
from scan2cap_amd.graphs import GraphedCallable
from scan2cap_amd.pipeline import GeometrySlots
bench, wl, model, dd, batch, msa, dev = T._setup("cfg3")
cfg = bench.LossConfig(msa)
from scan2cap_amd.optim import FusedAdam
opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)
slots = GeometrySlots(model.backbone_net, dd["point_clouds"], 1)
step = bench.make_step(model, wl, cfg, opt, None, dev)
def body():
    d = dict(dd); d["_geometry"] = slots.geometry(0); return step(d)
g = GraphedCallable(body).capture()
slots.refill(0, dd["point_clouds"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
out = []
for i in range(n):
    slots.acquire(0); l = g(); slots.release(0); slots.refill(0, dd["point_clouds"])
    if i % 20 == 0 or i == n - 1:
        out.append((i, float(l.detach())))
print(" ".join("%d:%.3f" % x for x in out))
bad = [k for k, v in model.state_dict().items() if v.is_floating_point() and not torch.isfinite(v).all()]
print("non-finite tensors:", bad)
