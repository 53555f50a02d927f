import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
import bench
from scan2cap_amd.loss_helper import get_scene_cap_loss
wl = bench.WORKLOADS["cfg3"]; dev = torch.device("cuda")
vocabulary, embeddings, table = bench.make_vocab(wl["V"])
msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
torch.manual_seed(0)
model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
dd0 = bench.to_device(bench.make_batch(wl, wl["B"], 42, table, msa), dev)
cfg = bench.LossConfig(msa)
step = bench.make_step(model, wl, cfg, opt, None, dev)
for _ in range(3): step(dd0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(dd0); torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
print("%-60s %6s %10s" % ("op", "count", "cuda_ms"))
for e in rows[:45]:
    print("%-60s %6d %10.3f" % (e.key[:60], e.count, e.device_time_total / 1e3))
