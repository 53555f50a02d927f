"""Micro-benchmark of the nine operators at the SA1..SA4 shapes of cfg3
(B=8, N=40000, C=132).  Times with HIP events on torch's current stream (the
stream the kernels are launched on)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from scan2cap_amd.pointnet2 import _ext
from scan2cap_amd.synthetic import scene_xyz


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--N", type=int, default=40000)
    ap.add_argument("--C", type=int, default=132)
    ap.add_argument("--mode", default="volume")
    a = ap.parse_args()
    B, N, C = a.B, a.N, a.C
    xyz = torch.from_numpy(scene_xyz(B, N, mode=a.mode)).cuda()
    feats = torch.randn(B, C, N, device="cuda")
    stages = [(2048, 0.2, 64, 128), (1024, 0.4, 32, 256), (512, 0.8, 16, 256), (256, 1.2, 16, 256)]
    cur_xyz, cur_feats = xyz, feats
    for (m, r, ns, cout) in stages:
        n = cur_xyz.shape[1]
        c = cur_feats.shape[1]
        t = timeit(lambda: _ext.furthest_point_sampling(cur_xyz, m), iters=3, warmup=1)
        print(f"fps            n={n:6d} m={m:5d}: {t:10.1f} us  ({t/(m-1):.3f} us/round)")
        inds = _ext.furthest_point_sampling(cur_xyz, m)
        xyz_t = cur_xyz.transpose(1, 2).contiguous()
        t = timeit(lambda: _ext.gather_points(xyz_t, inds))
        print(f"gather_points  n={n:6d} m={m:5d}: {t:10.1f} us")
        new_xyz = _ext.gather_points(xyz_t, inds).transpose(1, 2).contiguous()
        t = timeit(lambda: _ext.ball_query(new_xyz, cur_xyz, r, ns))
        tests = B * m * n
        print(f"ball_query     n={n:6d} m={m:5d} ns={ns}: {t:10.1f} us  ({tests/t/1e3:.1f} Gtests/s)")
        idx = _ext.ball_query(new_xyz, cur_xyz, r, ns)
        t = timeit(lambda: _ext.group_points(cur_feats, idx))
        byts = B * c * m * ns * 4 + B * m * ns * 4
        print(f"group_points   c={c:4d} m={m:5d} ns={ns}: {t:10.1f} us  ({byts/t/1e6:.2f} TB/s out+idx)")
        g = torch.randn(B, c, m, ns, device="cuda")
        t = timeit(lambda: _ext.group_points_grad(g, idx, n))
        print(f"group_pts_grad c={c:4d} m={m:5d} ns={ns}: {t:10.1f} us")
        cur_xyz = new_xyz
        cur_feats = torch.randn(B, cout, m, device="cuda")
    # FP stages
    for (n, m, c) in [(512, 256, 256), (1024, 512, 256)]:
        unk = torch.rand(B, n, 3, device="cuda")
        kn = torch.rand(B, m, 3, device="cuda")
        t = timeit(lambda: _ext.three_nn(unk, kn))
        print(f"three_nn       n={n} m={m}: {t:10.1f} us")
        d2, idx = _ext.three_nn(unk, kn)
        w = torch.rand(B, n, 3, device="cuda")
        f = torch.randn(B, c, m, device="cuda")
        t = timeit(lambda: _ext.three_interpolate(f, idx, w))
        print(f"three_interp   n={n} m={m} c={c}: {t:10.1f} us")
        g = torch.randn(B, c, n, device="cuda")
        t = timeit(lambda: _ext.three_interpolate_grad(g, idx, w, m))
        print(f"three_int_grad n={n} m={m} c={c}: {t:10.1f} us")


if __name__ == "__main__":
    main()
