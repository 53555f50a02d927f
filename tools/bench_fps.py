"""FPS at the SA1 shapes: wave-owned cells (4/8/16 waves) vs the bucket-list kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd.pointnet2 import _ext
from scan2cap_amd.synthetic import scene_xyz
from tools.bench_ops import timeit

for (B, N, m) in ((8, 40000, 2048), (16, 80000, 2048)):
    for mode in ("volume", "surface"):
        xyz = torch.from_numpy(scene_xyz(B, N, mode=mode)).cuda()
        res = []
        ref = None
        for impl, waves in (("cells", 16), ("cells", -2), ("cells", -3), ("cells", -4), ("cells", -16)):
            _ext.FPS_LARGE_IMPL, _ext.FPS_CELLS_WAVES = impl, waves
            out = _ext.furthest_point_sampling(xyz, m)
            if ref is None:
                ref = out
            assert torch.equal(out, ref), (impl, waves)
            t = timeit(lambda: _ext.furthest_point_sampling(xyz, m), iters=3, warmup=1)
            res.append("%s%s %7.0f us (%.2f us/pick)" % (impl, waves or "", t, t / (m - 1)))
        print("B=%d N=%d %-7s: %s" % (B, N, mode, " | ".join(res)))
