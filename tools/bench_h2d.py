"""Host->device rate of one cfg3 batch (the reference hands CapNet device tensors after
`data_dict[key].cuda()` in lib/solver.py; this is what a PCIe-inclusive step costs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench

wl = bench.WORKLOADS["cfg3"]
vocabulary, embeddings, table = bench.make_vocab(wl["V"])
msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
batch = bench.make_batch(wl, wl["B"], 42, table, msa)
host = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in batch.items()}
host = {k: v for k, v in host.items() if torch.is_tensor(v)}
nbytes = sum(v.numel() * v.element_size() for v in host.values())
for pinned in (False, True):
    src = {k: (v.pin_memory() if pinned else v) for k, v in host.items()}
    s = torch.cuda.Stream()
    for _ in range(2):
        with torch.cuda.stream(s):
            dst = {k: v.to("cuda", non_blocking=True) for k, v in src.items()}
        s.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        with torch.cuda.stream(s):
            dst = {k: v.to("cuda", non_blocking=True) for k, v in src.items()}
        s.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%s host memory: %.1f MB per batch in %.2f ms = %.1f GB/s" %
          ("pinned" if pinned else "pageable", nbytes / 1e6, dt * 1e3, nbytes / dt / 1e9))
