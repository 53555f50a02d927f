import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_configs_gpu as T
from tests import golden_common as gc
from scan2cap_amd.loss_helper import get_scene_cap_loss
from scan2cap_amd.opbyop import op_by_op
import contextlib
bench, wl, model, dd, batch, msa, dev = T._setup("cfg3")
cfg = bench.LossConfig(msa)
state = {k: v.clone() for k, v in model.state_dict().items()}
KEYS = ("sa1_features", "sa2_features", "sa3_features", "sa4_features", "fp2_features", "vote_xyz",
        "vote_features", "aggregated_vote_features", "_head_rows")
def run():
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    d = model(dict(dd), use_tf=True, is_eval=False)
    for k in KEYS:
        if d[k].requires_grad: d[k].retain_grad()
    d = get_scene_cap_loss(d, dev, cfg, None, detection=True, caption=True, orientation=False, distance=False)
    d["loss"].backward()
    return d
F = run()
with op_by_op(), gc.forced_vote_sampling(model, F["aggregated_vote_inds"]):
    O = run()
    with gc.ulp_noise(model, 5):
        N = run()
def st(a, b):
    a, b = a.double(), b.double()
    s = float(b.abs().max()); r = float(b.pow(2).mean().sqrt())
    d = a - b
    return "max %.2e rms %.2e nnz(>1e-3 max) %.4f" % (float(d.abs().max()) / s, float(d.pow(2).mean().sqrt()) / r,
                                          float((d.abs() > 1e-3 * s).float().mean()))
for k in KEYS:
    print("%-26s F-O: %s | N-O: %s" % (k, st(F[k], O[k]), st(N[k], O[k])))
for k in KEYS:
    if F[k].grad is not None and O[k].grad is not None:
        print("grad %-21s F-O: %s | N-O: %s" % (k, st(F[k].grad, O[k].grad), st(N[k].grad, O[k].grad)))
# ball query of the vote aggregation: same rows?
from scan2cap_amd.pointnet2 import _ext
def bq(d):
    return _ext.ball_query(d["aggregated_vote_xyz"].contiguous(), d["vote_xyz"].contiguous(), 0.3, 16)
a, b, c = bq(F), bq(O), bq(N)
print("vote ball-query rows differing: F-O %d  N-O %d of %d" % (int((a != b).any(-1).sum()), int((c != b).any(-1).sum()), a.shape[0] * a.shape[1]))
