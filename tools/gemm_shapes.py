"""Library GEMM calls (aten::mm / addmm / bmm / matmul) of one eager cfg3 train step,
grouped by input shapes, with their GPU time: which shapes the library handles badly."""
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
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, fused=True)
dd0 = bench.to_device(bench.make_batch(wl, wl["B"], 42, table, msa), dev)
cfg = bench.LossConfig(msa)


def step():
    dd = dict(dd0)
    opt.zero_grad(set_to_none=True)
    dd = model(dd, True, False)
    dd = get_scene_cap_loss(dd, dev, cfg, None)
    dd["loss"].backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm"):
        rows.append((e.device_time_total, e.count, e.key, str(e.input_shapes)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("library GEMMs: %.3f ms in %d calls" % (tot / 1e3, sum(r[1] for r in rows)))
for t, c, k, sh in rows[:40]:
    print("%8.1f us  x%-2d %-12s %s" % (t, c, k, sh))
