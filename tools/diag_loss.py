import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_configs_gpu as T
from scan2cap_amd import loss_helper as lh
from scan2cap_amd.models.proposal_module import ProposalModule
bench, wl, model, dd, batch, msa, dev = T._setup("cfg3")
cfg = bench.LossConfig(msa)
with torch.no_grad():
    F = model(dict(dd), use_tf=True, is_eval=False)
res = {}
for fused in (True, False):
    lh.FUSED_DETECTION_LOSS = fused
    d = dict(F)
    vote_xyz = F["vote_xyz"].detach().clone().requires_grad_(True)
    agg = F["aggregated_vote_xyz"].detach().clone().requires_grad_(True)
    rows = F["_head_rows"].detach().clone().transpose(1, 2).requires_grad_(True)   # (B,nout,K) like `net`
    d["vote_xyz"], d["aggregated_vote_xyz"] = vote_xyz, agg
    d = model.proposal.decode_scores(rows, d, 18, 1, 18, model.proposal.mean_size_arr)
    d = lh.get_scene_cap_loss(d, dev, cfg, None, detection=True, caption=False, orientation=False, distance=False)
    d["loss"].backward()
    res[fused] = (float(d["loss"]), vote_xyz.grad.clone(), agg.grad.clone(), rows.grad.clone())
lh.FUSED_DETECTION_LOSS = True
def st(a, b):
    a, b = a.double(), b.double()
    return "max|b| %.3e  max|a-b| %.3e  rel %.2e  nnz diff>1e-3max: %d" % (float(b.abs().max()), float((a - b).abs().max()),
            float((a - b).abs().max() / b.abs().max()), int(((a - b).abs() > 1e-3 * b.abs().max()).sum()))
print("loss", res[True][0], res[False][0])
for i, n in ((1, "d vote_xyz"), (2, "d aggregated_vote_xyz"), (3, "d head rows")):
    print(n, st(res[True][i], res[False][i]))
a, b = res[True][1], res[False][1]
bad = ((a - b).abs() > 1e-3 * b.abs().max()).any(-1)
print("rows of vote_xyz grad that differ:", int(bad.sum()), "of", bad.numel())
if bad.any():
    i = bad.nonzero()[:5]
    for bi, si in i.tolist():
        print(bi, si, a[bi, si].tolist(), b[bi, si].tolist())
