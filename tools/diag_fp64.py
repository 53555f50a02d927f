"""Diagnosis: float64 evaluation of the op-by-op model (torch ops only; index-producing ops
still see float32 xyz) as ground truth for the fused-fp32 and op-by-op-fp32 gradients.

    python tools/diag_fp64.py cfg3            (GPU)
    python tools/diag_fp64.py small --cpu     (oracle ops, debugging)
"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from tests import golden_common as gc
from scan2cap_amd.loss_helper import get_scene_cap_loss
from scan2cap_amd.opbyop import op_by_op
from scan2cap_amd.pointnet2 import pointnet2_utils as pu, _ext

cpu = "--cpu" in sys.argv
name = sys.argv[1]
if cpu:
    from oracle import torch_ext
    for n in torch_ext.NAMES:
        setattr(_ext, n, getattr(torch_ext, n))
wl = bench.WORKLOADS.get(name) or dict(B=2, N=4096, C=4, K=64, V=200, train=True, desc="small")
dev = torch.device("cpu" if cpu else "cuda")
vocabulary, embeddings, table = bench.make_vocab(wl["V"])
msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
torch.manual_seed(0)
model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).train()
dd = bench.to_device(bench.make_batch(wl, wl["B"], 42, table, msa), dev)
cfg = bench.LossConfig(msa)
state = {k: v.clone() for k, v in model.state_dict().items()}


def run(d_in):
    model.zero_grad(set_to_none=True)
    d = model(dict(d_in), use_tf=True, is_eval=False)
    d = get_scene_cap_loss(d, dev, cfg, None, detection=True, caption=True,
                           orientation=False, distance=False)
    d["loss"].backward()
    return d, {n: p.grad.detach().double().clone() for n, p in model.named_parameters()
               if p.grad is not None}


# --- dtype-generic replacements of the autograd wrappers -----------------------------
def fps(xyz, n):
    return _ext.furthest_point_sampling(xyz.float().contiguous(), n)

def gather_op(f, idx):
    return torch.gather(f, 2, idx.long().unsqueeze(1).expand(-1, f.shape[1], -1))

def three_nn(u, k):
    _, idx = _ext.three_nn(u.float().contiguous(), k.float().contiguous())
    nb = torch.gather(k.unsqueeze(1).expand(-1, u.shape[1], -1, -1), 2,
                      idx.long().unsqueeze(-1).expand(-1, -1, -1, 3))
    return ((u.unsqueeze(2) - nb) ** 2).sum(-1).sqrt(), idx

def three_interp(f, idx, w):
    B, C, m = f.shape
    n = idx.shape[1]
    g = torch.gather(f, 2, idx.long().view(B, 1, n * 3).expand(-1, C, -1)).view(B, C, n, 3)
    return (g * w.unsqueeze(1)).sum(-1)

def group_op(f, idx):
    B, C, N = f.shape
    _, m, ns = idx.shape
    return torch.gather(f, 2, idx.long().view(B, 1, m * ns).expand(-1, C, -1)).view(B, C, m, ns)

def ball_q(radius, ns, xyz, new_xyz):
    return _ext.ball_query(new_xyz.float().contiguous(), xyz.float().contiguous(), radius, ns)


if not cpu:
    got, g_f = run(dd)                       # fused fp32
    forced = got["aggregated_vote_inds"]
    model.load_state_dict(state)
with op_by_op():
    if cpu:
        got, g_o = run(dd)
        forced = got["aggregated_vote_inds"]
    else:
        with gc.forced_vote_sampling(model, forced):
            want, g_o = run(dd)              # op-by-op fp32
    model.load_state_dict(state)
    saved = {k: getattr(pu, k) for k in ("furthest_point_sample", "gather_operation", "three_nn",
                                          "three_interpolate", "grouping_operation", "ball_query")}
    pu.furthest_point_sample, pu.gather_operation, pu.three_nn = fps, gather_op, three_nn
    pu.three_interpolate, pu.grouping_operation, pu.ball_query = three_interp, group_op, ball_q
    model.double()
    torch.set_default_dtype(torch.float64)
    dd64 = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v)
            for k, v in dd.items()}
    with gc.forced_vote_sampling(model, forced):
        truth, g_t = run(dd64)
    torch.set_default_dtype(torch.float32)
    for k, v in saved.items():
        setattr(pu, k, v)
print("loss fp64 %.9f  opbyop %.9f" % (float(truth["loss"]), float((got if cpu else want)["loss"])),
      "" if cpu else "fused %.9f" % float(got["loss"]))
print("%-60s %10s %12s %12s" % ("param", "max|truth|", "err opbyop", "err fused"))
for n in g_t:
    t = g_t[n]
    s = max(1e-30, float(t.abs().max()))
    eo = float((g_o[n] - t).abs().max()) / s
    ef = float((g_f[n] - t).abs().max()) / s if not cpu else float("nan")
    if max(eo, ef if ef == ef else 0) > 1e-4:
        print("%-60s %10.3e %12.3e %12.3e" % (n, s, eo, ef))
