import numpy as np, torch, sys
sys.path.insert(0, '.')
from tests import variants_common as vc
from scan2cap_amd.models import graph_module as gm
gold = np.load('tests/golden/variants.npz')
for name in ("edge_add", "graph_conv"):
  for qk in (True, False):
    gm.USE_QUERY_KERNEL = qk
    m = gm.GraphModule(**vc.GRAPH_DIMS, **vc.GRAPH_CASES[name]).eval()
    vc.fill_params(m, seed=17); m = m.cuda()
    dd = {k: torch.from_numpy(v.copy()).cuda() for k, v in vc.graph_inputs().items()}
    with torch.no_grad(): dd = m(dd)
    for k in vc.GRAPH_OUT_KEYS:
        want = gold["graph/%s/%s" % (name, k)]; got = dd[k].cpu().numpy().astype(np.float64)
        d = np.abs(got - want).reshape(want.shape[0], -1).max(1)
        print(name, "kernel" if qk else "torch", k, d)
