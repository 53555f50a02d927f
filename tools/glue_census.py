"""Which Python lines of the cfg3 train step issue framework (ATen) ops other than GEMMs --
the "torch glue" rows of profiles/*_kernel_stats.csv.  TorchDispatchMode over one eager step;
call site = innermost frame inside this repo (custom Function.backward frames included)."""
import collections, os, sys, traceback
import torch
from torch.utils._python_dispatch import TorchDispatchMode
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_configs_gpu as T
from scan2cap_amd.loss_helper import get_scene_cap_loss

SKIP = ("view", "reshape", "transpose", "permute", "expand", "slice", "select", "unsqueeze",
        "squeeze", "detach", "alias", "as_strided", "t.default", "empty", "_unsafe_view",
        "size", "stride", "is_", "unbind", "split", "chunk", "narrow", "numel", "dim",
        "_local_scalar", "lift_fresh", "_to_copy", "result_type", "empty_like", "new_empty",
        "mm.default", "bmm", "addmm", "matmul", "unflatten", "flatten", "movedim")

class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.c = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if ("scan2cap_amd" in fr.filename or fr.filename.endswith("bench.py")) \
                        and "glue_census" not in fr.filename:
                    where = "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
                    break
            if where == "?":          # issued by the autograd engine itself: name the node it is running
                try:
                    node = torch._C._current_autograd_node()
                    where = "engine:" + (node.name() if node is not None else "-")
                except Exception:
                    pass
            big = max([a.numel() for a in args if torch.is_tensor(a)] + [0])
            self.c[(name.replace("aten.", ""), where, big)] += 1
        return func(*args, **(kwargs or {}))

bench, wl, model, dd, batch, msa, dev = T._setup("cfg3")
cfg = bench.LossConfig(msa)
from scan2cap_amd.optim import FusedAdam
opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)
step = bench.make_step(model, wl, cfg, opt, None, dev)
geo = model.backbone_net.compute_geometry(dd["point_clouds"])
d0 = dict(dd); d0["_geometry"] = geo
for _ in range(2):
    step(d0)
torch.cuda.synchronize()
cz = Census()
with cz:
    step(d0)
torch.cuda.synchronize()
tot = 0
for (name, where, big), n in sorted(cz.c.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    tot += n
    print("%3d  %-34s %-32s max numel %d" % (n, name, where, big))
print("total", tot)
