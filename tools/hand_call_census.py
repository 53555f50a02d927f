"""Every C-ABI call of one eager cfg3 train step with its leading integer arguments (the
shape) and its HIP-event time, aggregated per (entry point, shape): which hand kernels run
at which sizes.  python tools/hand_call_census.py [name-filter]"""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_configs_gpu as T
from scan2cap_amd import _C
from scan2cap_amd.loss_helper import get_scene_cap_loss

flt = sys.argv[1] if len(sys.argv) > 1 else ""
bench, wl, model, dd, batch, msa, dev = T._setup("cfg3")
cfg = bench.LossConfig(msa)
log, active = [], {"on": False}
orig = _C.call

def call(name, *args, allow=()):
    if not active["on"] or flt not in name:
        return orig(name, *args, allow=allow)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = orig(name, *args, allow=allow)
    e1.record()
    ints = []
    for a in args:
        if isinstance(a, bool) or not isinstance(a, int) or a > (1 << 31):
            break
        ints.append(a)
    log.append((name, tuple(ints), e0, e1))
    return rc
_C.call = call

def step():
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
for name, ints, e0, e1 in log:
    c = agg.setdefault((name, ints), [0, 0.0])
    c[0] += 1; c[1] += e0.elapsed_time(e1) * 1e3
tot = 0
for (name, ints), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += us
    print("%8.1f us %4d x %-34s %s" % (us, cnt, name, ints))
print("total %.1f us over %d calls (an event pair adds ~3 us per call)" % (tot, len(log)))
