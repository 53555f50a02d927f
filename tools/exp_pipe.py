import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from scan2cap_amd.pipeline import GeometryPipeline
wl = bench.WORKLOADS["cfg2"]; dev = torch.device("cuda")
vocabulary, embeddings, table = bench.make_vocab(wl["V"])
msa = np.random.Generator(np.random.PCG64(5)).uniform(0.3, 1.5, size=(18, 3))
model = bench.build_model(wl, vocabulary, embeddings, msa).to(dev).eval()
dd = bench.to_device(bench.make_batch(wl, wl["B"], 42, table, msa), dev)
pc = dd["point_clouds"]
bb = model.backbone_net
def geo_time(nstreams, R=9):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    for s in streams:
        with torch.cuda.stream(s): bb.compute_geometry(pc)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(R):
        with torch.cuda.stream(streams[i % nstreams]): bb.compute_geometry(pc)
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("geometry only, %d stream(s): %.2f ms per batch (host submit %.2f ms)" % (nstreams, dt / R * 1e3, host / R * 1e3))
for n in (1, 2, 3): geo_time(n)

# ---- replica of the bench loop with a timeline ----
from collections import deque
from scan2cap_amd.pipeline import flatten_geometry, unflatten_geometry
from scan2cap_amd.graphs import GraphedCallable
depth = 3
pipe = GeometryPipeline(bb, depth=depth)
geo0 = bb.compute_geometry(pc)
static_geo = [torch.empty_like(t) for t in flatten_geometry(geo0)]
for d_, s_ in zip(static_geo, flatten_geometry(geo0)): d_.copy_(s_)
dd["_geometry"] = unflatten_geometry(static_geo)
def fwd():
    with torch.no_grad():
        return model(dict(dd), use_tf=False, is_eval=True)["objectness_scores"]
g = GraphedCallable(fwd).capture()
ev = lambda: torch.cuda.Event(enable_timing=True)
timeline = []
def submit():
    side = pipe.streams[pipe._next % depth]; pipe._next += 1
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        s = ev(); s.record(side)
        geo = bb.compute_geometry(pc)
        e = ev(); e.record(side)
    return geo, s, e
q = deque(submit() for _ in range(depth))
origin = ev(); origin.record()
for i in range(12):
    geo, gs, ge = q.popleft()
    q.append(submit())
    torch.cuda.current_stream().wait_event(ge)
    rs = ev(); rs.record()
    for d_, s_ in zip(static_geo, flatten_geometry(geo)): d_.copy_(s_, non_blocking=True)
    g()
    re_ = ev(); re_.record()
    timeline.append((gs, ge, rs, re_))
torch.cuda.synchronize()
for i, (gs, ge, rs, re_) in enumerate(timeline):
    print("step %2d: geo [%7.2f -> %7.2f]  replay [%7.2f -> %7.2f]" % (
        i, origin.elapsed_time(gs), origin.elapsed_time(ge), origin.elapsed_time(rs), origin.elapsed_time(re_)))
