import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_configs_gpu as T
from scan2cap_amd.graphs import GraphedCallable
from scan2cap_amd.pipeline import GeometrySlots
bench, wl, model, dd, batch, msa, dev = T._setup("cfg3")
cfg = bench.LossConfig(msa)
def new_opt():
    return torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, capturable=True, fused=True)
state = {k: v.clone() for k, v in model.state_dict().items()}
eager = bench.make_step(model, wl, cfg, new_opt(), None, dev)
ref = [float(eager(dd).detach())]
w1 = {k: v.clone() for k, v in model.state_dict().items()}
ref.append(float(eager(dd).detach()))
# eager again from the same state: run-to-run spread of the eager path itself
model.load_state_dict(state)
eager2 = bench.make_step(model, wl, cfg, new_opt(), None, dev)
ref2 = [float(eager2(dd).detach()) for _ in range(2)]
model.load_state_dict(state)
opt = new_opt()
step = bench.make_step(model, wl, cfg, opt, None, dev)
slots = GeometrySlots(model.backbone_net, dd["point_clouds"], 1)
def body():
    d = dict(dd); d["_geometry"] = slots.geometry(0); return step(d)
g = GraphedCallable(body).capture()
model.load_state_dict(state)
for st in opt.state.values():
    for v in st.values():
        if torch.is_tensor(v): v.zero_()
slots.refill(0, dd["point_clouds"])
slots.acquire(0); l0 = float(g().detach()); slots.release(0); slots.refill(0, dd["point_clouds"])
torch.cuda.synchronize()
wg = {k: v.clone() for k, v in model.state_dict().items()}
slots.acquire(0); l1 = float(g().detach()); torch.cuda.synchronize()
print("eager", ref, "eager again", ref2, "graph", [l0, l1])
rows = []
for k in w1:
    if w1[k].is_floating_point() and "running" not in k and "num_batches" not in k:
        d = (w1[k] - wg[k]).abs()
        rows.append((float((d > 1e-4).float().mean()), float(d.max()), k))
rows.sort(reverse=True)
print("weights after one update, graph vs eager: fraction of entries > 0.1 lr apart / max:")
for r in rows[:8]: print("  %.3f %.2e %s" % r)
