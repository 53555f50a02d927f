"""Which python lines issue the framework glue ops (copy_, fill_, zero_, cat, small elementwise)
of one eager cfg3 train step (TorchDispatchMode + traceback)."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench
wl = bench.WORKLOADS["cfg3"]; dev = torch.device("cuda")
vocabulary, embeddings, table = bench.make_vocab(wl["V"])
msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
torch.manual_seed(0)
model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, fused=True)
dd0 = bench.to_device(bench.make_batch(wl, wl["B"], 42, table, msa), dev)
cfg = bench.LossConfig(msa)
step = bench.make_step(model, wl, cfg, opt, None, dev)
for _ in range(2): step(dd0)
WATCH = ("copy_", "fill_", "zero_", "cat", "clone", "mul", "add", "gather", "sum", "div", "gt", "scatter_add_", "zeros", "contiguous")
agg = collections.Counter()
class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            big = [a for a in args if torch.is_tensor(a) and a.is_cuda]
            if big:
                st = traceback.extract_stack()
                site = next((f for f in reversed(st) if "scan2cap_amd/" in f.filename), None)
                if site is None:
                    site = next((f for f in reversed(st) if "bench.py" in f.filename), None)
                key = ("%s:%d" % (site.filename.split("repo/")[-1], site.lineno) if site else "?", name)
                agg[key] += 1
        return func(*args, **(kwargs or {}))
with Spy():
    step(dd0)
torch.cuda.synchronize()
for (site, name), n in agg.most_common(60):
    print("%3d %-14s %s" % (n, name, site))
