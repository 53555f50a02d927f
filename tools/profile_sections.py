"""Section-level timing of one cfg3 train step with HIP events (torch current
stream): where the milliseconds go between the hand-written kernels."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from scan2cap_amd.loss_helper import get_scene_cap_loss

def main():
    wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
    dev = torch.device("cuda")
    vocabulary, embeddings, table = bench.make_vocab(wl["V"])
    msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
    torch.manual_seed(0)
    model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
    dd0 = bench.to_device(bench.make_batch(wl, wl["B"], 42, table, msa), dev)
    cfg = bench.LossConfig(msa)
    marks = []
    def mark(name):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
    def step(timed):
        dd = dict(dd0)
        if timed: mark("start")
        opt.zero_grad(set_to_none=False)
        bb = model.backbone_net
        xyz, feats = bb._break_up_pc(dd["point_clouds"])
        if timed: mark("break_up_pc")
        for i in (1, 2, 3, 4):
            xyz, feats, inds = getattr(bb, "sa%d" % i)(xyz, feats)
            dd["sa%d_xyz" % i] = xyz; dd["sa%d_features" % i] = feats; dd["sa%d_inds" % i] = inds
            if timed: mark("sa%d" % i)
        f = bb.fp1(dd["sa3_xyz"], dd["sa4_xyz"], dd["sa3_features"], dd["sa4_features"])
        f = bb.fp2(dd["sa2_xyz"], dd["sa3_xyz"], dd["sa2_features"], f)
        dd["fp2_features"] = f; dd["fp2_xyz"] = dd["sa2_xyz"]; dd["fp2_inds"] = dd["sa1_inds"][:, :1024]
        if timed: mark("fp1+fp2")
        xyz = dd["fp2_xyz"]; dd["seed_inds"] = dd["fp2_inds"]; dd["seed_xyz"] = xyz; dd["seed_features"] = f
        xyz, f = model.vgen(xyz, f)
        f = f.div(torch.norm(f, p=2, dim=1).unsqueeze(1))
        dd["vote_xyz"] = xyz; dd["vote_features"] = f
        if timed: mark("vgen")
        dd = model.proposal(xyz, f, dd)
        if timed: mark("proposal")
        dd = model.graph(dd)
        if timed: mark("graph")
        dd = model.caption(dd, True, False)
        if timed: mark("caption")
        dd = get_scene_cap_loss(dd, dev, cfg, None)
        if timed: mark("loss")
        dd["loss"].backward()
        if timed: mark("backward")
        opt.step()
        if timed: mark("adam")
    for _ in range(3): step(False)
    torch.cuda.synchronize()
    acc = {}
    R = 5
    for _ in range(R):
        marks.clear(); step(True); torch.cuda.synchronize()
        for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
            acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
    tot = sum(acc.values())
    for k, v in acc.items():
        print("%-12s %8.3f ms  %5.1f%%" % (k, v / R, 100 * v / tot))
    print("%-12s %8.3f ms" % ("total", tot / R))

if __name__ == "__main__":
    main()
