"""Evaluation-mode CapNet (greedy decode of every proposal, models/caption_module.py
`_forward_scene_batch`) at the cfg3 / cfg5 shapes: ms per batch, peak memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 256
wl = dict(bench.WORKLOADS["cfg3"]); wl.update(B=B, N=N, K=K)
dev = torch.device("cuda")
vocabulary, embeddings, table = bench.make_vocab(wl["V"])
msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
torch.manual_seed(0)
model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).eval()
dd0 = bench.to_device(bench.make_batch(wl, B, 42, table, msa), dev)
def run():
    with torch.no_grad():
        return model(dict(dd0), use_tf=False, is_eval=True)
out = run(); torch.cuda.synchronize()
print({k: tuple(out[k].shape) for k in ("lang_cap", "topdown_attn", "valid_masks")})
torch.cuda.reset_peak_memory_stats()
t0 = time.perf_counter()
R = 3
for _ in range(R): run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / R
print("eval forward B=%d N=%d K=%d: %.1f ms/batch = %.1f scenes/s, peak mem %.1f GB" % (
    B, N, K, dt * 1e3, B / dt, torch.cuda.max_memory_allocated() / 2**30))
