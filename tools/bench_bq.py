"""Ball query at the SA1 shape: grid kernel (csrc/s2c_bq_grid.hip) vs brute force, both
synthetic modes, cfg3 and cfg5 sizes.  HIP events on the launch stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scan2cap_amd.pointnet2 import _ext
from scan2cap_amd.synthetic import scene_xyz
from tools.bench_ops import timeit

for (B, N, m, r, ns) in ((8, 40000, 2048, 0.2, 64), (16, 80000, 2048, 0.2, 64), (8, 8192, 1024, 0.4, 32)):
    for mode in ("volume", "surface"):
        xyz = torch.from_numpy(scene_xyz(B, N, mode=mode)).cuda()
        inds = _ext.furthest_point_sampling(xyz, m)
        new_xyz = torch.gather(xyz, 1, inds.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
        a = _ext.ball_query(new_xyz, xyz, r, ns)
        b = _ext.ball_query_bruteforce(new_xyz, xyz, r, ns)
        assert torch.equal(a, b)
        t_grid = timeit(lambda: _ext.ball_query(new_xyz, xyz, r, ns), iters=20, warmup=3)
        t_bf = timeit(lambda: _ext.ball_query_bruteforce(new_xyz, xyz, r, ns), iters=5, warmup=1)
        ab = 4 * (3 * B * N + 3 * B * m + B * m * ns)
        print("B=%d N=%d m=%d r=%.1f ns=%d %-7s: grid %7.1f us (%.0f GB/s alg)   brute %7.1f us"
              % (B, N, m, r, ns, mode, t_grid, ab / t_grid / 1e3, t_bf))
