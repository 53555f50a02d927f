"""Gradients of one train step (tests/test_train_loop_gpu.py set-up) with the streaming GEMM
on vs off: which parameters move."""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_train_loop_gpu as T
from scan2cap_amd import _C
from scan2cap_amd.loss_helper import get_scene_cap_loss
bench, wl, model, opt, dd, cfg, dev = T._setup()
lib = _C.load()
def grads(on):
    lib.s2c_gemm_set_stream(on)
    model.zero_grad(set_to_none=True)
    d = model(dict(dd), use_tf=True, is_eval=False)
    d = get_scene_cap_loss(d, dev, cfg, None)
    d["loss"].backward()
    return float(d["loss"]), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, \
        {k: d[k].detach().clone() for k in ("vote_xyz", "aggregated_vote_xyz", "objectness_scores", "center") if k in d}
state = {k: v.clone() for k, v in model.state_dict().items()}
l1, g1, o1 = grads(1)
model.load_state_dict(state)
l0, g0, o0 = grads(0)
model.load_state_dict(state)
l0b, g0b, o0b = grads(0)
print("loss on %.6f off %.6f off-again %.6f" % (l1, l0, l0b))
for k in o1:
    print("out %-24s on-off %.3e   off-off %.3e  (scale %.2e)" % (k, float((o1[k] - o0[k]).abs().max()), float((o0b[k] - o0[k]).abs().max()), float(o0[k].abs().max())))
rows = []
for n in g0:
    s = max(1e-12, float(g0[n].abs().max()))
    rows.append((float((g1[n] - g0[n]).abs().max()) / s, float((g0b[n] - g0[n]).abs().max()) / s, n, s))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%-60s on-off %.2e  off-off %.2e  scale %.2e" % (r[2], r[0], r[1], r[3]))
