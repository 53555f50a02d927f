"""Which GEMMs of the cfg3 train step (or, with an argument, of another workload's forward) still go to the library (torch.mm / addmm / bmm / matmul /
F.linear -> hipBLASLt), with shapes, call sites and HIP-event times (eager step)."""
import collections, os, sys, traceback
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_configs_gpu as T
from scan2cap_amd.loss_helper import get_scene_cap_loss

WL = sys.argv[1] if len(sys.argv) > 1 else "cfg3"     # cfg3 (train step) | cfg2 | cfg3e | cfg5 (forward)
bench, wl, model, dd, batch, msa, dev = T._setup(WL)
cfg = bench.LossConfig(msa)
log = []
names = ["mm", "addmm", "bmm", "matmul", "baddbmm"]
orig = {n: getattr(torch, n) for n in names}
orig_lin = torch.nn.functional.linear
active = {"on": False}

def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "scan2cap_amd" in fr.filename:
            return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    return "?"

def wrap(name, fn):
    def f(*a, **k):
        if not active["on"]:
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        shapes = tuple(tuple(x.shape) for x in a if torch.is_tensor(x))
        log.append((name, shapes, site(), e0, e1))
        return out
    return f
for n in names:
    setattr(torch, n, wrap(n, orig[n]))
torch.nn.functional.linear = wrap("linear", orig_lin)

def step():
    if not wl["train"]:
        with torch.no_grad():
            model(dict(dd), use_tf=False, is_eval=True)
        return
    model.zero_grad(set_to_none=True)
    d = model(dict(dd), use_tf=True, is_eval=False)
    d = get_scene_cap_loss(d, dev, cfg, None, detection=True, caption=True, orientation=False, distance=False)
    d["loss"].backward()
for _ in range(2):
    step()
active["on"] = True
step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, shapes, where, e0, e1 in log:
    k = (name, shapes, where)
    t = e0.elapsed_time(e1) * 1e3
    c = agg.setdefault(k, [0, 0.0])
    c[0] += 1; c[1] += t
tot = 0
for (name, shapes, where), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += us
    print("%8.1f us %3d x %-8s %-28s %s" % (us, cnt, name, where, shapes))
print("total %.1f us over %d calls (event pairs add ~3 us per call)" % (tot, len(log)))
