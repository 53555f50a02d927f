"""Per-section kernel-launch counts and GPU time of one eager cfg3 train step
(torch.profiler), to find where the small-kernel swarm comes from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
import bench
from scan2cap_amd.loss_helper import get_scene_cap_loss

wl = bench.WORKLOADS["cfg3"]
dev = torch.device("cuda")
vocabulary, embeddings, table = bench.make_vocab(wl["V"])
msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
torch.manual_seed(0)
model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).train()
from scan2cap_amd.optim import FusedAdam
opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)
dd0 = bench.to_device(bench.make_batch(wl, wl["B"], 42, table, msa), dev)
cfg = bench.LossConfig(msa)

HIST = {}

def sections():
    dd = dict(dd0)
    out = {}
    def run(name, fn):
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            r = fn()
            torch.cuda.synchronize()
        evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
        out[name] = (len(evs), sum(e.device_time for e in evs) / 1e3)
        hist = {}
        for e in evs:
            k = e.name[:90]
            c = hist.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += e.device_time / 1e3
        HIST[name] = hist
        return r
    opt.zero_grad(set_to_none=True)      # as the bench step: fresh gradient tensors, no accumulate kernels
    run("backbone", lambda: model.backbone_net(dd))
    def vg():
        xyz = dd["fp2_xyz"]; f = dd["fp2_features"]
        dd["seed_inds"] = dd["fp2_inds"]; dd["seed_xyz"] = xyz; dd["seed_features"] = f
        xyz, f = model.vgen(xyz, f)
        f = f.div(torch.norm(f, p=2, dim=1).unsqueeze(1))
        dd["vote_xyz"] = xyz; dd["vote_features"] = f
    run("vgen+norm", vg)
    run("proposal", lambda: model.proposal(dd["vote_xyz"], dd["vote_features"], dd))
    run("graph", lambda: model.graph(dd))
    run("caption", lambda: model.caption(dd, True, False))
    run("loss", lambda: get_scene_cap_loss(dd, dev, cfg, None))
    run("backward", lambda: dd["loss"].backward())
    run("adam", lambda: opt.step())
    return out

for _ in range(2):
    sections()
for k, (n, ms) in sections().items():
    print("%-10s %5d launches %8.3f ms" % (k, n, ms))

for sec in ("graph", "caption", "loss", "proposal", "backward"):
    print("==", sec)
    for k, (c, ms) in sorted(HIST[sec].items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %4d x %8.3f ms  %s" % (c, ms, k))
